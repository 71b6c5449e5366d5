#!/usr/bin/env python3
"""Per-wave phase cycles of the PAIR step launch (two envs per two-wave workgroup, split roles) in the bench.py state mix.
DIAGNOSTIC build: make -C trex-gym_amd/csrc variant XFLAGS="-DTREX_STAMPS=1 -DTREX_PAIR_LAUNCH=1" SUFFIX=_pstamps; TREX_LIB selects it.
Wave 0 of a workgroup (even wave index) runs the tree dynamics of both envs, wave 1 the contact generation of both; the waits
at the two workgroup barriers of a substep have their own slots. Shares, not lengths, are meaningful (the stamps fence the
scheduler, add global read-modify-writes and, in this form, cost the build 24 spilled registers)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "trex-gym_amd"))
from trex_gym import _capi, sharding  # noqa: E402

SLOTS = [(0, "FK (own env)"), (16, "wait at barrier A"), (1, "contact generation (both envs; wave 1)"),
         (17, "tree dynamics (both envs; wave 0)"), (18, "wait at barrier B"), (4, "base factor read, vel update"),
         (5, "row walks"), (6, "z0 stash + B build"), (7, "PGS sweeps"), (8, "results + integrate")]
NS = 20


def main():
    dev = torch.device("cuda:0")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    pre = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    m = _capi.Model()
    b = _capi.Batch(m, n)
    obs = torch.zeros(n, 75, device=dev); rew = torch.zeros(n, device=dev); done = torch.zeros(n, dtype=torch.uint8, device=dev)
    ids = torch.arange(n, device=dev)
    lo, hi = torch.tensor(m.lower, dtype=torch.float32, device=dev), torch.tensor(m.upper, dtype=torch.float32, device=dev)
    phase = (ids * 1000) // n
    b.reset(obs)
    for t in range(pre):
        b.step(sharding.synthetic_actions(ids, t, lo, hi, seed=0, device=dev), obs, rew, done)
        mk = (phase == ((-(t + 1)) % 1000)).to(torch.uint8)
        if bool(mk.any()):
            b.reset(obs, mk)
    dbg = torch.zeros(4096 + NS * n, device=dev)
    b.debug_step(sharding.synthetic_actions(ids, pre, lo, hi, seed=0, device=dev), obs, dbg)
    torch.cuda.synchronize()
    cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    b.contact_stats(cnt, None)
    full = dbg.cpu().numpy()[4096:4096 + NS * n].reshape(NS, n)
    env_of_wave = full[11].astype(np.int64)
    c = cnt.cpu().numpy()[env_of_wave]
    d = {k: full[k].copy() for k, _ in SLOTS}
    d[1] = full[1] + full[9] + full[10] + full[12] + full[13]      # contact generation with its sub-stamps
    tot = sum(d.values())
    print("envs %d (pair launch: %d workgroups), contacts per env mean %.2f max %d" % (n, n // 2, c.mean(), c.max()))
    for role, name in ((0, "wave 0 (tree dynamics of both envs; runs the HEAVY env)"), (1, "wave 1 (contacts of both envs; runs the LIGHT env)")):
        sel = (np.arange(n) & 1) == role
        t = tot[sel]
        print("%s: own env's contacts mean %.2f; cycles mean %.3g p99 %.3g max %.3g" % (name, c[sel].mean(), t.mean(), np.percentile(t, 99), t.max()))
        slow = np.argsort(t)[-max(1, sel.sum() // 100):]
        for k, nm in SLOTS:
            v = d[k][sel]
            print("    %-42s %10.0f %6.1f %%   slowest 1 %%: %10.0f" % (nm, v.mean(), 100 * v.mean() / t.mean(), v[slow].mean()))
    # the critical path of a workgroup per substep = FK, max(contacts, tree), the per-env tail
    w0, w1 = (np.arange(n) & 1) == 0, (np.arange(n) & 1) == 1
    print("between the barriers: tree dynamics %.3g against contact generation %.3g cycles per launch (mean over workgroups); the longer of the two %.3g" % (
        d[17][w0].mean(), d[1][w1].mean(), np.maximum(d[17][w0], d[1][w1]).mean()))
    tail = d[4] + d[5] + d[6] + d[7] + d[8]
    print("per-env tail (rows .. integrate): heavy wave %.3g, light wave %.3g; end-of-launch idle of the light wave = difference" % (tail[w0].mean(), tail[w1].mean()))


if __name__ == "__main__":
    main()
