#!/usr/bin/env python3
"""What does a concurrent memory-moving kernel on another stream cost the step kernel? (stand-in for the RCCL
all-gather of an 8-GPU run, which cannot be measured on a one-GPU box): per step, a device-to-device copy of
`mb` MB is launched on a side stream beside the next step.
    python scripts/overlap_probe.py [mb ...]"""
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "trex-gym_amd"))
from trex_gym import sharding  # noqa: E402
from trex_gym.vec_env import TrexVecEnv  # noqa: E402

dev = torch.device("cuda:0")
n = 4096
env = TrexVecEnv(n, device=dev, max_episode_steps=1000)
ids = torch.arange(n, device=dev)
lo, hi = env.model.lower, env.model.upper
pool = torch.stack([sharding.synthetic_actions(ids, t, lo, hi, seed=0, device=dev) for t in range(16)])
env.reset_tensor()
env.set_episode_steps(((ids * 1000) // n).to(torch.int32))
for t in range(1000):
    env.step_tensor(pool[t % 16])
side = torch.cuda.Stream()
for mb in [0.0] + [float(x) for x in (sys.argv[1:] or ["1.26", "10", "40"])]:
    words = int(mb * 1e6 / 4)
    a = torch.zeros(max(words, 1), device=dev)
    b = torch.zeros(max(words, 1), device=dev)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(200):
            env.step_tensor(pool[t % 16])
            if mb > 0:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    b.copy_(a)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 200
    print("side-stream copy of %5.2f MB per step: %.4f ms/step (%.3f M env-steps/s)" % (mb, dt * 1e3, n / dt / 1e6), flush=True)
