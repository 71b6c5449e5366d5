#!/usr/bin/env python3
"""Per-wave phase cycles of the one-env-per-wave step kernel in the bench.py state mix (DIAGNOSTIC build:
make -C trex-gym_amd/csrc stamps; TREX_LIB selects it; balanced by contact rank like the product launch). Every wave accumulates s_memtime deltas
per phase over the 5 substeps of ONE launch; the script prints the mean per phase, the same for the slowest
waves, and wave time against contact count. Shares, not lengths, are meaningful (the stamps fence the
scheduler and add global read-modify-writes)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "trex-gym_amd"))
from trex_gym import _capi, sharding  # noqa: E402

NAMES = ["FK", "contact generation", "velocities, inertia, bias", "ABA pass 2 (LDS)", "base inverse, pass 3, vel update",
         "row walks", "z0 stash + B build", "PGS sweeps", "results + integrate"]


def main():
    dev = torch.device("cuda:0")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    pre = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    cycle = int(sys.argv[3]) if len(sys.argv) > 3 else 0     # 16: bench.py's actions (16 draws per env, cycled); 0: fresh draws
    m = _capi.Model()
    b = _capi.Batch(m, n)
    obs = torch.zeros(n, 75, device=dev); rew = torch.zeros(n, device=dev); done = torch.zeros(n, dtype=torch.uint8, device=dev)
    ids = torch.arange(n, device=dev)
    lo, hi = torch.tensor(m.lower, dtype=torch.float32, device=dev), torch.tensor(m.upper, dtype=torch.float32, device=dev)
    # staggered episodes like bench.py: env i is reset when (t + phase_i) % 1000 == 0
    phase = (ids * 1000) // n
    b.reset(obs)
    for t in range(pre):
        b.step(sharding.synthetic_actions(ids, t % cycle if cycle else t, lo, hi, seed=0, device=dev), obs, rew, done)
        mk = (phase == ((-(t + 1)) % 1000)).to(torch.uint8)
        if bool(mk.any()):
            b.reset(obs, mk)
    dbg = torch.zeros(4096 + 16 * n, device=dev)
    b.debug_step(sharding.synthetic_actions(ids, pre % cycle if cycle else pre, lo, hi, seed=0, device=dev), obs, dbg)
    torch.cuda.synchronize()
    cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    b.contact_stats(cnt, None)
    full = dbg.cpu().numpy()[4096:4096 + 16 * n].reshape(16, n)
    raw = full[:14]
    alive, lamnz = full[14], full[15]    # point slots alive / holding an impulse at the start of a sweep, summed over the 300 sweeps
    env_of_wave = raw[11].astype(np.int64)          # the launch is balanced like the product's: wave k runs env perm[k]
    c = cnt.cpu().numpy()[env_of_wave]              # contacts of the env each wave ran
    d = raw[:9].copy()
    sub = raw[9:11]                      # contact generation split: [broad phase + table, scan]; d[1] = the rest
    sel = raw[12:14] if raw.shape[0] >= 14 else np.zeros((2, n))   # [deepest vertices + set-up, fill + passes] (K >= 2 path)
    selp = sel
    d[1] += sub.sum(0) + sel.sum(0)
    tot = d.sum(0)
    print("waves %d, contacts per env mean %.2f max %d" % (n, c.mean(), c.max()))
    print("wave cycles: mean %.3g  median %.3g  p90 %.3g  p99 %.3g  max %.3g  (max/mean %.2f)" % (
        tot.mean(), np.median(tot), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(), tot.max() / tot.mean()))
    slow = np.argsort(tot)[-max(1, n // 100):]
    print("%-36s %12s %7s %14s" % ("phase", "mean cycles", "share", "slowest 1 %"))
    for k, name in enumerate(NAMES):
        print("%-36s %12.0f %6.1f %% %14.0f" % (name, d[k].mean(), 100 * d[k].mean() / tot.mean(), d[k][slow].mean()))
    rest = d[1] - sub.sum(0) - sel.sum(0)
    print("contact generation = broad phase + table %.0f + scan %.0f + deepest vertices, ranking, set-up %.0f + candidate fill, passes %.0f + output %.0f" % (
        sub[0].mean(), sub[1].mean(), sel[0].mean(), sel[1].mean(), rest.mean()))
    for lo_, hi_ in ((0, 0), (1, 4), (5, 8), (9, 13)):
        sel = (c >= lo_) & (c <= hi_)
        if sel.any():
            rows = 25 + 3 * c[sel].mean()
            print("envs with %2d..%2d contacts: %5d  wave cycles mean %.3g max %.3g; sweeps mean %.3g = %.1f cycles per row visit" % (
                lo_, hi_, sel.sum(), tot[sel].mean(), tot[sel].max(), d[7][sel].mean(), d[7][sel].mean() / (300 * rows)))
            if c[sel].sum() > 0:
                print("    of their point slots, alive at the start of a sweep: %.1f %%; holding a normal impulse: %.1f %%" % (
                    100 * alive[sel].sum() / (300.0 * c[sel].sum()), 100 * lamnz[sel].sum() / (300.0 * c[sel].sum())))


    # how well do the balance keys predict a wave's work? least squares of the wave cycles on (contacts) and on
    # (contacts, live points per sweep); meaningful where every wave is alone on its SIMD (<= 1024 envs)
    L = alive / 300.0
    for name, cols in (("contacts", [c]), ("contacts + live points per sweep", [c, L])):
        A = np.stack([np.ones(n)] + [np.asarray(x, dtype=np.float64) for x in cols], 1)
        coef, *_ = np.linalg.lstsq(A, tot, rcond=None)
        res = tot - A @ coef
        print("fit of the wave cycles on %s: coefficients %s, residual std %.3g (of std %.3g), worst %.3g" % (
            name, " ".join("%.3g" % v for v in coef), res.std(), tot.std(), np.abs(res).max()))
    # (workgroup k shares its SIMD with k +- 1024 ...)
    if n == 4096:
        order = np.argsort(tot)[::-1][:8]
        # joints on a stop after the step (their limit rows were probably active: compiled rows + a column read from LDS)
        q = obs[:, :25].cpu().numpy()
        lo_h, hi_h = np.asarray(m.lower)[None, :], np.asarray(m.upper)[None, :]
        on_stop = ((q <= lo_h + 1e-6) | (q >= hi_h - 1e-6)).sum(1)[env_of_wave]
        print("slowest waves: cycles, contacts, joints on a stop | sweeps, contact gen, pass 2, B build | SIMD mates' (cycles, contacts, on a stop)")
        for w in order:
            mates = [(w + 1024 * k) % 4096 for k in (1, 2, 3)]
            print("  %.3g %2d %2d | %.3g %.3g %.3g %.3g | %s" % (tot[w], c[w], on_stop[w], d[7][w], d[1][w], d[3][w], d[6][w],
                  " ".join("(%.3g, %d, %d)" % (tot[m_], c[m_], on_stop[m_]) for m_ in mates)))
            print("      contact gen: table %.3g scan %.3g set-up %.3g fill+passes %.3g output %.3g; live points per sweep %.2f" % (
                sub[0][w], sub[1][w], selp[0][w], selp[1][w], d[1][w] - sub[:, w].sum() - selp[:, w].sum(), alive[w] / 300.0))
        for k in range(0, 6):
            sel = on_stop == k if k < 5 else on_stop >= k
            if sel.any():
                print("envs with %s%d joints on a stop: %5d  wave cycles mean %.3g; sweeps mean %.3g" % (">=" if k == 5 else "", k, sel.sum(), tot[sel].mean(), d[7][sel].mean()))
        simd = tot.reshape(4, 1024)
        print("per-SIMD (k mod 1024): sum of its 4 waves' cycles mean %.3g max %.3g; max of its 4 waves mean %.3g max %.3g" % (
            simd.sum(0).mean(), simd.sum(0).max(), simd.max(0).mean(), simd.max(0).max()))


if __name__ == "__main__":
    main()
