#!/usr/bin/env python3
"""Study (CPU, numpy): the 25 motor rows of a projected-Gauss-Seidel sweep as ONE linear map of the sweep's input.

While no motor row changes its clamp status, the block of 25 dependent row visits
    for j: d_j = a_j ? y_j : 0;  y += B[:, j] d_j           (a_j: row j strictly inside its torque bounds)
is linear in the y it starts from:  raw = R y_m,  y' = y + C y_m  (R 25 x 25 unit lower triangular, C rows x 25), and the
clamp is verified afterwards on `raw` (one vector compare). This script takes the constraint systems of real substeps from
the f64 oracle (a study build of oracle/trex_oracle.c with -DORACLE_ROWS_HOOK), and runs 60 sweeps three ways:
    f64 rows   - the reference
    f32 rows   - what the kernel does today (row by row)
    f32 matrix - sweep 0 by rows, then the matrix form with the check; a failed check = that sweep by rows + a rebuild
and reports (a) how often the check fails per substep, (b) the error of both f32 forms against f64.

    python3 scripts/matrix_sweep_proto.py [envs] [steps]
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from oracle import oracle as orc, trex_model  # noqa: E402

f32 = np.float32


def study_lib():
    out = os.path.join(tempfile.gettempdir(), "liboracle_rows_hook.so")
    src = os.path.join(ROOT, "oracle", "trex_oracle.c")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", "-ffp-contract=off", "-DORACLE_ROWS_HOOK",
                           "-o", out, src, "-lm"])
    return out


def sweeps_rows(B, rhs, lo, hi, kind, fof, mu, iters, dt):
    """row by row, in the arithmetic of dt (np.float64 / np.float32)"""
    nr = len(rhs)
    B = B.astype(dt); y = rhs.astype(dt).copy(); lam = np.zeros(nr, dt)
    lo = lo.astype(dt); hi = hi.astype(dt); mu = dt(mu)
    for _ in range(iters):
        for r in range(nr):
            l, h = lo[r], hi[r]
            if fof[r] >= 0:
                h = mu * lam[fof[r]]; l = -h
            d = min(max(y[r], l - lam[r]), h - lam[r])
            if d != 0:
                lam[r] += d
                y += B[:, r] * d
    return lam, y


def build_CR(B, m0, act):
    """f32, the streaming recurrence of the kernel: T_k = response of the block to e_k; R[j][k] = T_k[row j] at its visit"""
    nr = B.shape[0]
    nm = len(act)
    Cm = np.zeros((nr, nm), f32)          # C[:, k]
    R = np.eye(nm, dtype=f32)             # R[j, k]
    for j in range(nm):
        if not act[j]:
            continue
        b = B[:, m0 + j]
        bl = B[m0:m0 + nm, m0 + j].copy()
        bl[:j + 1] = 0
        for k in range(j + 1):
            s = R[j, k]
            if s != 0:
                Cm[:, k] += b * s
                R[:, k] += bl * s
    return Cm, R


def sweeps_matrix(B, rhs, lo, hi, kind, fof, mu, iters, max_fail=3, first=1, predict=True):
    nr = len(rhs)
    B = B.astype(f32); y = rhs.astype(f32).copy(); lam = np.zeros(nr, f32)
    lo = lo.astype(f32); hi = hi.astype(f32); mu = f32(mu)
    mot = np.nonzero(kind == 1)[0]
    m0, nm = int(mot[0]), len(mot)
    assert (mot == np.arange(m0, m0 + nm)).all()
    fails = 0
    flips = 0
    mat_sweeps = 0
    builds = 0
    Cm = R = act = None

    def row(r):
        nonlocal y
        l, h = lo[r], hi[r]
        if fof[r] >= 0:
            h = mu * lam[fof[r]]; l = -h
        d = min(max(y[r], l - lam[r]), h - lam[r])
        if d != 0:
            lam[r] += d
            y += B[:, r] * d

    for it in range(iters):
        for r in range(0, m0):
            row(r)
        done = False
        if it >= first and fails < max_fail:
            if Cm is None:
                # the rows taken as unclamped: inside their bounds now (predict: and still inside with the residual they hold)
                z = lam[m0:m0 + nm] + (y[m0:m0 + nm] if predict else f32(0))
                act = (z > lo[m0:m0 + nm]) & (z < hi[m0:m0 + nm])
                Cm, R = build_CR(B, m0, act)
                builds += 1
            ym = y[m0:m0 + nm].copy()
            raw = np.zeros(nm, f32)
            for k in range(nm):
                raw += R[:, k] * ym[k]
            blo = lo[m0:m0 + nm] - lam[m0:m0 + nm]; bhi = hi[m0:m0 + nm] - lam[m0:m0 + nm]
            dexp = np.where(act, raw, f32(0))
            if (np.minimum(np.maximum(raw, blo), bhi) == dexp).all():
                yn = y.copy()
                for k in range(nm):
                    yn += Cm[:, k] * ym[k]
                y = yn
                lam[m0:m0 + nm] += dexp
                done = True
                mat_sweeps += 1
            else:
                fails += 1
        if not done:
            for r in range(m0, m0 + nm):
                row(r)
            Cm = None
        for r in range(m0 + nm, nr):
            row(r)
    return lam, y, fails, builds, mat_sweeps


def main():
    n_env = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    n_step = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    first = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    predict = bool(int(sys.argv[5])) if len(sys.argv) > 5 else True
    model = trex_model.compile_model(orc.default_asset_urdf())
    lib = C.CDLL(study_lib())
    systems = []
    HOOK = C.CFUNCTYPE(None, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double)

    grab = {"on": False}

    def hook(nr, Bp, rp, fr):
        if not grab["on"]:
            return
        B = np.ctypeslib.as_array(Bp, (nr * nr,)).reshape(nr, nr).copy()
        rd = np.ctypeslib.as_array(rp, (nr * 6,)).reshape(nr, 6).copy()
        systems.append((B, rd, fr))

    cb = HOOK(hook)
    # rebuild an Oracle on the study library
    import oracle.oracle as om
    real_build = om.build
    om.build = lambda force=False: [lib._name, lib._name]
    o = orc.Oracle(model)
    om.build = real_build
    o.lib.oracle_set_rows_hook(cb)
    rng = np.random.default_rng(0)
    lo_q, hi_q = np.asarray(model["q_lower"])[1:], np.asarray(model["q_upper"])[1:]
    for e in range(n_env):
        s = o.new_state()
        o.reset(s)
        for t in range(n_step):
            grab["on"] = (t % every == every - 1)
            o.step(s, rng.uniform(lo_q, hi_q))      # (the order of the entries does not matter for a uniform draw)
    grab["on"] = False
    print("%d substep systems captured (%d envs x %d steps, every %d)" % (len(systems), n_env, n_step, every))
    iters = 60
    hist = np.zeros(8, int)
    e_rows, e_mat = [], []
    tot_mat = tot_sw = tot_flips = 0
    nrows = []
    for (B, rd, fr) in systems:
        rhs, lo, hi, kind, fof = rd[:, 0], rd[:, 1], rd[:, 2], rd[:, 3].astype(int), rd[:, 4].astype(int)
        if not (kind == 1).any():
            continue
        l64, y64 = sweeps_rows(B, rhs, lo, hi, kind, fof, fr, iters, np.float64)
        l32, y32 = sweeps_rows(B, rhs, lo, hi, kind, fof, fr, iters, np.float32)
        lm, ym, fails, flips, ms = sweeps_matrix(B, rhs, lo, hi, kind, fof, fr, iters, first=first, predict=predict)
        mot = kind == 1
        ref = B[mot] @ l64                    # ~ joint velocity change per unit diag
        sc = np.abs(ref).max() + 1e-30
        e_rows.append(np.abs(B[mot] @ l32.astype(np.float64) - ref).max() / sc)
        e_mat.append(np.abs(B[mot] @ lm.astype(np.float64) - ref).max() / sc)
        hist[min(fails, 7)] += 1
        tot_mat += ms; tot_sw += iters; tot_flips += flips
        nrows.append(len(rhs))
    e_rows, e_mat = np.array(e_rows), np.array(e_mat)
    print("rows per system: mean %.1f max %d" % (np.mean(nrows), max(nrows)))
    print("failed checks per substep (0..7+):", hist.tolist(), " matrix sweeps %.1f %% of all" % (100.0 * tot_mat / max(tot_sw, 1)))
    print("builds of (C, R) per substep: %.2f" % (tot_flips / max(1, len(nrows))))
    for name, e in (("f32 rows", e_rows), ("f32 matrix", e_mat)):
        print("%-10s error of B_m lam against f64, relative to its largest entry: median %.2e  p90 %.2e  p99 %.2e  max %.2e"
              % (name, np.median(e), np.quantile(e, 0.9), np.quantile(e, 0.99), e.max()))
    print("ratio matrix / rows: median %.2f  p90 %.2f  max %.2f" % (np.median(e_mat / np.maximum(e_rows, 1e-12)),
          np.quantile(e_mat / np.maximum(e_rows, 1e-12), 0.9), (e_mat / np.maximum(e_rows, 1e-12)).max()))


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "lazy"):
    main()


# ---------------------------------------------------------------------------------------------------------------------
# Form Z: C only (registers), no R. y' = y + C y_m per sweep; the impulses of a WINDOW of sweeps are recovered afterwards from
# the sum of the window's inputs by ONE forward substitution (lam += R ysum, R never formed); whether a clamp could have
# bound inside the window is decided per sweep by a cheap SUFFICIENT test with u >= row sums of |R| (u = (I - |L_a|)^-1 1):
#   active row j     |lam0_j| + u_j G < hi_j          G = sum over the window's sweeps of max_k |y_k| (k active)
#   saturated row j  sigma_j y_j - (u_j - 1) ymax > 0
# A failed test = flush the window, that sweep by rows (always right), the set of unclamped rows re-derived, C rebuilt if it changed.
COST = dict(rows=130, mat=70, flush=100, build=750)


def sweeps_lazy(B, rhs, lo, hi, kind, fof, mu, iters, first=2, max_builds=3, max_fail=6, exact_u=False):
    nr = len(rhs)
    B = B.astype(f32); y = rhs.astype(f32).copy(); lam = np.zeros(nr, f32)
    lo = lo.astype(f32); hi = hi.astype(f32); mu = f32(mu)
    mot = np.nonzero(kind == 1)[0]
    m0, nm = int(mot[0]), len(mot)
    ms = slice(m0, m0 + nm)
    Bmm = B[ms, ms]
    slots = 0
    builds = fails = mat = 0
    Cm = act = u = None
    ysum = np.zeros(nm, f32); G = f32(0); lam0 = None
    giveup = False

    def row(r):
        nonlocal y
        l, h = lo[r], hi[r]
        if fof[r] >= 0:
            h = mu * lam[fof[r]]; l = -h
        d = min(max(y[r], l - lam[r]), h - lam[r])
        if d != 0:
            lam[r] += d
            y += B[:, r] * d

    def flush():
        nonlocal ysum, G, slots
        if not ysum.any():
            return
        raw = ysum.copy()
        for j in range(nm):           # forward substitution, the rows taken as unclamped pass their value on
            if act[j]:
                raw[j + 1:] += Bmm[j + 1:, j] * raw[j]
        lam[ms] += np.where(act, raw, f32(0))
        ysum = np.zeros(nm, f32); G = f32(0)
        slots += COST["flush"]

    def build():
        nonlocal Cm, u, builds, slots, act
        z = lam[ms] + y[ms]
        act = (z > lo[ms]) & (z < hi[ms])
        T = np.zeros((nr, nm), f32)
        T[ms, :] = np.eye(nm, dtype=f32)
        Rrows = np.zeros((nm, nm), f32)
        for j in range(nm):
            Rrows[j] = T[m0 + j]
            if act[j]:
                s = T[m0 + j].copy()                 # row j of R
                T += np.outer(B[:, m0 + j], s).astype(f32)
        Cm = T.copy(); Cm[ms, :] -= np.eye(nm, dtype=f32)
        if exact_u:
            u = np.abs(Rrows).sum(1).astype(f32)
        else:
            u = np.ones(nm, f32)
            for j in range(nm):
                if act[j]:
                    u[j + 1:] += np.abs(Bmm[j + 1:, j]) * u[j]
        builds += 1
        slots += COST["build"]

    for it in range(iters):
        for r in range(0, m0):
            row(r)
        done = False
        if it >= first and not giveup:
            if Cm is None:
                build()
                lam0 = lam[ms].copy()
            ym = y[ms].copy()
            ymax = np.abs(ym[act]).max() if act.any() else f32(0)
            Gn = G + ymax
            sig = np.where(lam0 > 0, f32(1), f32(-1))
            ok_act = np.abs(lam0) + u * Gn < hi[ms]
            ok_sat = sig * ym - (u - f32(1)) * ymax > 0
            if np.where(act, ok_act, ok_sat).all():
                yn = y.copy()
                for k in range(nm):
                    yn += Cm[:, k] * ym[k]
                y = yn
                ysum += ym; G = Gn
                mat += 1; slots += COST["mat"]
                done = True
            else:
                fails += 1
                flush()
        if not done:
            for r in range(m0, m0 + nm):
                row(r)
            slots += COST["rows"]
            if Cm is not None:
                z = lam[ms] + y[ms]
                new_act = (z > lo[ms]) & (z < hi[ms])
                if (new_act != act).any():
                    Cm = None
                    if builds >= max_builds:
                        giveup = True
                else:
                    lam0 = lam[ms].copy()
                if fails >= max_fail:
                    giveup = True
        for r in range(m0 + nm, nr):
            row(r)
    if Cm is not None:
        flush()
    return lam, y, slots, builds, fails, mat


def main_lazy():
    n_env = int(sys.argv[2]); n_step = int(sys.argv[3]); every = int(sys.argv[4])
    first = int(sys.argv[5]) if len(sys.argv) > 5 else 2
    exact_u = bool(int(sys.argv[6])) if len(sys.argv) > 6 else False
    systems = capture(n_env, n_step, every)
    iters = 60
    S, Bd, Fl, Mt, e_rows, e_mat = [], [], [], [], [], []
    for (B, rd, fr) in systems:
        rhs, lo, hi, kind, fof = rd[:, 0], rd[:, 1], rd[:, 2], rd[:, 3].astype(int), rd[:, 4].astype(int)
        if not (kind == 1).any():
            continue
        l64, _ = sweeps_rows(B, rhs, lo, hi, kind, fof, fr, iters, np.float64)
        l32, _ = sweeps_rows(B, rhs, lo, hi, kind, fof, fr, iters, np.float32)
        lm, _, slots, builds, fails, mat = sweeps_lazy(B, rhs, lo, hi, kind, fof, fr, iters, first=first, exact_u=exact_u)
        mot = kind == 1
        ref = B[mot] @ l64
        sc = np.abs(ref).max() + 1e-30
        e_rows.append(np.abs(B[mot] @ l32.astype(np.float64) - ref).max() / sc)
        e_mat.append(np.abs(B[mot] @ lm.astype(np.float64) - ref).max() / sc)
        S.append(slots); Bd.append(builds); Fl.append(fails); Mt.append(mat)
    S = np.array(S); e_rows = np.array(e_rows); e_mat = np.array(e_mat)
    base = iters * COST["rows"]
    print("%d systems; motor-block slots per substep: mean %.0f (rows: %d) = %.1f %%; p10 %.0f p50 %.0f p90 %.0f; worse than rows: %.1f %%"
          % (len(S), S.mean(), base, 100 * S.mean() / base, np.quantile(S, .1), np.median(S), np.quantile(S, .9), 100 * (S > base).mean()))
    print("builds per substep %.2f; failed tests %.2f; matrix sweeps %.1f of %d" % (np.mean(Bd), np.mean(Fl), np.mean(Mt), iters))
    for name, e in (("f32 rows", e_rows), ("f32 lazy", e_mat)):
        print("%-10s error against f64: median %.2e  p90 %.2e  p99 %.2e  max %.2e" % (name, np.median(e), np.quantile(e, 0.9), np.quantile(e, 0.99), e.max()))


def capture(n_env, n_step, every):
    model = trex_model.compile_model(orc.default_asset_urdf())
    lib = C.CDLL(study_lib())
    systems = []
    HOOK = C.CFUNCTYPE(None, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double)
    grab = {"on": False}

    def hook(nr, Bp, rp, fr):
        if grab["on"]:
            systems.append((np.ctypeslib.as_array(Bp, (nr * nr,)).reshape(nr, nr).copy(),
                            np.ctypeslib.as_array(rp, (nr * 6,)).reshape(nr, 6).copy(), fr))

    cb = HOOK(hook)
    import oracle.oracle as om
    real_build = om.build
    om.build = lambda force=False: [lib._name, lib._name]
    o = orc.Oracle(model)
    om.build = real_build
    o.lib.oracle_set_rows_hook(cb)
    rng = np.random.default_rng(0)
    lo_q, hi_q = np.asarray(model["q_lower"])[1:], np.asarray(model["q_upper"])[1:]
    for e in range(n_env):
        s = o.new_state()
        o.reset(s)
        for t in range(n_step):
            grab["on"] = (t % every == every - 1)
            o.step(s, rng.uniform(lo_q, hi_q))
    grab["on"] = False
    capture.keep = (cb, o)
    return systems


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "lazy":
    main_lazy()
