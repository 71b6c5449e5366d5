#!/usr/bin/env python3
"""Soak test on the GPU box: N envs, uniform random actions (changing every 10 steps), thousands of steps;
counts containment events (non-finite env states) and prints state ranges."""
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "trex-gym_amd"))
from trex_gym import sharding  # noqa: E402
from trex_gym.vec_env import TrexVecEnv  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    n = 4096
    v = TrexVecEnv(n, device="cuda:0", collision=sys.argv[2] if len(sys.argv) > 2 else "hulls")
    v.reset_tensor()
    ids = torch.arange(n, device="cuda:0")
    events = torch.zeros((), dtype=torch.int64, device="cuda:0")
    rmin = torch.full((), 0.0, device="cuda:0")
    for t in range(steps):
        if t % 10 == 0:
            a = sharding.synthetic_actions(ids, t // 10, v.model.lower, v.model.upper, device="cuda:0")
        obs, rew, done = v.step_tensor(a)
        events += done.sum()
        rmin = torch.minimum(rmin, rew.min())
        if (t + 1) % 1000 == 0:
            st = v.get_state()
            print("step %6d  containment events so far %d  |qd| max %.1f  base z [%.2f, %.2f]  |xy| max %.1f  min reward %.3g  finite %s"
                  % (t + 1, events.item(), obs[:, 25:50].abs().max().item(), st[:, 2].min().item(), st[:, 2].max().item(),
                     st[:, :2].abs().max().item(), rmin.item(), bool(torch.isfinite(obs).all().item())), flush=True)


if __name__ == "__main__":
    main()
