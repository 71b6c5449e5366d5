#!/usr/bin/env python3
"""BASELINE config 3 on one GPU: env-steps/s of the PPO loop (4096 envs, nsteps 32, 32 minibatches) - rollout alone
(policy kernel + env step + statistics kernel per step) and including the learner, eager and with HIP-graph replay.
    python scripts/ppo_rate.py [noptepochs] [num_envs]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "trex-gym_amd"))
from trex_gym.ppo import PPO
from trex_gym.trex_train import build_environment

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
for graphs in (False, True):
    env = build_environment(n)
    agent = PPO(env, nsteps=32, nminibatches=32, noptepochs=epochs, seed=0, use_graphs=graphs)
    for _ in range(3):
        agent.update(agent.collect())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s0 = agent.total_env_steps
    tr = 0.0
    for _ in range(10):
        t1 = time.perf_counter()
        b = agent.collect()
        torch.cuda.synchronize()
        tr += time.perf_counter() - t1
        agent.update(b)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = agent.total_env_steps - s0
    print("PPO %d envs, %d epochs x 32 minibatches, graphs=%d: %.3f M env-steps/s including the learner; rollout alone %.3f M; "
          "update %.1f ms per iteration (%.4f ms per minibatch step)"
          % (n, epochs, graphs, steps / dt / 1e6, steps / tr / 1e6, (dt - tr) / 10 * 1e3, (dt - tr) / 10 / (epochs * 32) * 1e3), flush=True)
