#!/usr/bin/env python3
"""Which motor rows sit at their torque bound in the bench.py state mix (obs[:, 50:75] = the motor torque of the last
substep): share per joint, rows at the bound per env, and how stable the set is from one step to the next."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "trex-gym_amd"))
from trex_gym import _capi, sharding  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    n, pre = 4096, 600
    m = _capi.Model()
    b = _capi.Batch(m, n)
    obs = torch.zeros(n, 75, device=dev); rew = torch.zeros(n, device=dev); done = torch.zeros(n, dtype=torch.uint8, device=dev)
    ids = torch.arange(n, device=dev)
    lo, hi = torch.tensor(m.lower, dtype=torch.float32, device=dev), torch.tensor(m.upper, dtype=torch.float32, device=dev)
    phase = (ids * 1000) // n
    b.reset(obs)
    prev = None
    same, changed = 0, 0
    for t in range(pre):
        b.step(sharding.synthetic_actions(ids, t, lo, hi, seed=0, device=dev), obs, rew, done)
        mk = (phase == ((-(t + 1)) % 1000)).to(torch.uint8)
        if bool(mk.any()):
            b.reset(obs, mk)
        if t >= pre - 50:
            tau = obs[:, 50:75].abs()
            sat = tau >= 0.999 * tau.max()
            if prev is not None:
                same += int((sat == prev).all(1).sum()); changed += int((sat != prev).any(1).sum())
            prev = sat
    tau = obs[:, 50:75].abs().cpu().numpy()
    top = tau.max()
    sat = tau >= 0.999 * top
    print("largest |motor torque| %.4g; rows at it: %.2f %% of %d x 25" % (top, 100 * sat.mean(), n))
    print("share per joint (sorted-name order):", " ".join("%.0f" % (100 * x) for x in sat.mean(0)))
    cnt = sat.sum(1)
    print("rows at the bound per env: " + " ".join("%d:%.1f%%" % (k, 100 * (cnt == k).mean()) for k in range(0, 10)))
    print("envs whose set of rows at the bound is the same as one step earlier: %.1f %%" % (100.0 * same / max(1, same + changed)))
    for chunk in (5, 4, 3):
        groups = [list(range(i, min(25, i + chunk))) for i in range(0, 25, chunk)]
        clean = np.mean([[not sat[e, g].any() for g in groups] for e in range(n)])
        print("chunks of %d rows without a row at the bound: %.1f %%" % (chunk, 100 * clean))


if __name__ == "__main__":
    main()
