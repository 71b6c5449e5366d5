#!/usr/bin/env python3
"""How long does each wave of a step launch live, and why? Diagnostic build (make stamps): every wave
writes its s_memtime duration and the contact counts of its two envs. Bench scenario (uniform random
actions), 4096 envs, with and without the contact-count pairing (TREX_DEBUG_PAIR=1)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
os.environ.setdefault("TREX_LIB", os.path.join(ROOT, "trex-gym_amd", "trex_gym", "libtrex_hip_stamps.so"))
sys.path.insert(0, os.path.join(ROOT, "trex-gym_amd"))
from trex_gym import _capi, sharding  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    n = 4096
    m = _capi.Model()
    b = _capi.Batch(m, n)
    obs = torch.zeros(n, 75, device=dev); rew = torch.zeros(n, device=dev); done = torch.zeros(n, dtype=torch.uint8, device=dev)
    b.reset(obs)
    lo = torch.tensor(m.lower, dtype=torch.float32, device=dev); hi = torch.tensor(m.upper, dtype=torch.float32, device=dev)
    ids = torch.arange(n, device=dev)
    for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 230):
        a = sharding.synthetic_actions(ids, t, lo, hi, device=dev)
        b.step(a, obs, rew, done)
    dbg = torch.zeros(12 * 4096, device=dev)
    a = sharding.synthetic_actions(ids, 1000, lo, hi, device=dev)
    b.debug_step(a, obs, dbg)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy()
    cyc = d[4096:4096 + n // 2]
    code = d[8192:8192 + n // 2].astype(int)
    n0, n1 = code % 100, code // 100
    cg = d[12288:12288 + n // 2]
    cg1, cg2 = d[16384:16384 + n // 2], d[20480:20480 + n // 2]
    print("  of which small-hull scan mean %.3g max %.3g, large-hull scan mean %.3g max %.3g, point selection (K >= 2 re-scans, fast path) mean %.3g max %.3g"
          % (cg1.mean(), cg1.max(), cg2.mean(), cg2.max(), (cg - cg1 - cg2).mean(), (cg - cg1 - cg2).max()))
    print("contact generation per wave: mean %.3g  p90 %.3g  max %.3g cycles; corr(wave cycles, contact-generation cycles) %.2f"
          % (cg.mean(), np.percentile(cg, 90), cg.max(), np.corrcoef(cyc, cg)[0, 1]))
    tot = n0 + n1
    print("pairing in this launch:", "on" if os.environ.get("TREX_DEBUG_PAIR") else "off")
    print("wave cycles: mean %.3g  median %.3g  p90 %.3g  p99 %.3g  max %.3g  (max/mean %.2f)"
          % (cyc.mean(), np.median(cyc), np.percentile(cyc, 90), np.percentile(cyc, 99), cyc.max(), cyc.max() / cyc.mean()))
    print("contacts per wave (sum of the two envs): mean %.1f max %d" % (tot.mean(), tot.max()))
    for lo_, hi_ in ((0, 0), (1, 4), (5, 8), (9, 12), (13, 16), (17, 26)):
        sel = (tot >= lo_) & (tot <= hi_)
        if sel.any():
            print("  waves with %2d..%2d contacts: %4d  cycles mean %.3g  max %.3g" % (lo_, hi_, sel.sum(), cyc[sel].mean(), cyc[sel].max()))
    if not os.environ.get("TREX_DEBUG_PAIR"):   # wave w = envs 2w, 2w+1: what else makes a wave slow?
        st = torch.zeros(n, b.state_width, device=dev)
        b.get_state(st)
        z = st[:, 2].cpu().numpy().reshape(-1, 2).min(1)
        sp = st[:, 38:63].abs().max(1).values.cpu().numpy().reshape(-1, 2).max(1)
        zero = tot == 0
        print("zero-contact waves: corr(cycles, min base z) %.2f  corr(cycles, max |qd|) %.2f" % (
            np.corrcoef(cyc[zero], z[zero])[0, 1], np.corrcoef(cyc[zero], sp[zero])[0, 1]))
        for zl, zh in ((0, 1.0), (1.0, 2.0), (2.0, 3.0), (3.0, 9.0)):
            sel = zero & (z >= zl) & (z < zh)
            if sel.any():
                print("  zero-contact waves with min base z in [%.1f, %.1f): %4d  cycles mean %.3g max %.3g" % (zl, zh, sel.sum(), cyc[sel].mean(), cyc[sel].max()))
    ph = np.stack([d[28672 + 4096 * i: 28672 + 4096 * i + n // 2] for i in range(5)], 1)
    names = ["tree phases", "contact generation", "chain walk + row build", "solver sweeps", "results, integrate, epilogue FK"]
    print("per-wave phase cycles (mean over the waves | mean over the slowest 5 %% of the waves):")
    slow = cyc >= np.percentile(cyc, 95)
    for i, nm in enumerate(names):
        print("  %-32s %9.3g (%4.1f %%) | %9.3g (%4.1f %%)" % (nm, ph[:, i].mean(), 100 * ph[:, i].mean() / ph.sum(1).mean(),
                                                            ph[slow, i].mean(), 100 * ph[slow, i].mean() / ph[slow].sum(1).mean()))
    sel_t = cg - cg1 - cg2
    cnt = d[24576:24576 + n // 2].astype(np.int64)
    nb_, np_, nt_ = cnt % 1000, (cnt // 1000) % 1000, cnt // 1000000
    print("pass B per wave-step: body iterations mean %.1f max %d, passes mean %.1f max %d, candidate trips mean %.1f max %d" % (
        nb_.mean(), nb_.max(), np_.mean(), np_.max(), nt_.mean(), nt_.max()))
    A = np.stack([nb_, np_, nt_, np.ones_like(nb_)], 1).astype(float)
    coef = np.linalg.lstsq(A, sel_t, rcond=None)[0]
    print("least squares: selection cycles = %.0f per body + %.0f per pass + %.0f per trip + %.0f" % tuple(coef))
    kk = np.argsort(-sel_t)[:8]
    print("waves with the longest point selection (selection cycles, contacts env a, env b):",
          [(int(sel_t[i]), int(n0[i]), int(n1[i])) for i in kk])
    for lo_, hi_ in ((0, 0), (1, 3), (4, 4), (5, 8), (9, 13)):
        selw = (np.maximum(n0, n1) >= lo_) & (np.maximum(n0, n1) <= hi_)
        if selw.any():
            print("  waves whose heavier env has %d..%d contacts: selection mean %.3g max %.3g" % (lo_, hi_, sel_t[selw].mean(), sel_t[selw].max()))
    k = np.argsort(-cyc)[:8]
    print("slowest waves (cycles, of which contact generation, contacts env a, env b):",
          [(int(cyc[i]), int(cg[i]), int(n0[i]), int(n1[i])) for i in k])


if __name__ == "__main__":
    main()
