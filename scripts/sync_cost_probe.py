#!/usr/bin/env python3
"""What do the cross-stream dependencies of the pipelined row gather cost a chain of step launches? (one GPU; the
collective is stood in for by a 1.26 MB device-to-device copy on a side stream). Variants, per step t:
  none      step launches back to back
  fork      side stream waits for step t, copies its rows                      (what launching a collective does)
  fork+join ... and the compute stream waits for the copy of step t-1 before step t+1   (PipelinedGather, two row blocks)
  fork+join2/3  the compute stream waits for the copy of step t-2 / t-3 instead          (three / four row blocks)
  fork+hostjoin no wait on the compute stream; the HOST checks that copy t-2 has completed before it launches step t+1
    python scripts/sync_cost_probe.py"""
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
cur = torch.cuda.current_stream()
dst = torch.zeros_like(env.rows)
K = 300
for mode in ("none", "fork", "fork+join", "fork+join2", "fork+join3", "fork+hostjoin", "none"):
    fork_ev = [torch.cuda.Event() for _ in range(K)]
    done_ev = [torch.cuda.Event() for _ in range(K)]
    lag = {"fork+join": 1, "fork+join2": 2, "fork+join3": 3}.get(mode, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(K):
        env.step_tensor(pool[t % 16])
        if mode != "none":
            fork_ev[t].record(cur)
            side.wait_event(fork_ev[t])
            with torch.cuda.stream(side):
                dst.copy_(env.rows)
            done_ev[t].record(side)
            if lag and t >= lag:
                cur.wait_event(done_ev[t - lag])
            if mode == "fork+hostjoin" and t >= 2:
                done_ev[t - 2].synchronize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("%-14s %.4f ms/step (%.3f M env-steps/s)" % (mode, dt * 1e3, n / dt / 1e6), flush=True)
