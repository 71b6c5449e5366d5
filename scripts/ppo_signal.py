#!/usr/bin/env python3
"""Learning-signal experiment: PPO at the reference's hyper-parameters, mean reward per update.
    python scripts/ppo_signal.py [num_envs] [nsteps] [updates] [noptepochs] [graphs] [episode_steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "trex-gym_amd"))
from trex_gym.ppo import PPO
from trex_gym.trex_train import build_environment

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 32
updates = int(sys.argv[3]) if len(sys.argv) > 3 else 30
epochs = int(sys.argv[4]) if len(sys.argv) > 4 else 32
graphs = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False
episode = int(sys.argv[6]) if len(sys.argv) > 6 else 1000
env = build_environment(n, max_episode_steps=episode)
agent = PPO(env, nsteps=nsteps, nminibatches=32, noptepochs=epochs, lam=0.95, gamma=0.99, lr=3e-4, cliprange=0.2, ent_coef=0.0,
            seed=0, use_graphs=graphs)
t0 = time.perf_counter()
for it in range(updates):
    b = agent.collect()
    info = agent.update(b)
    print("it %2d reward/step %10.4f pg %8.4f vf %8.4f ent %7.3f  %.1fs" % (it, b[-1], info["policy_loss"], info["value_loss"],
                                                                             info["entropy"], time.perf_counter() - t0), flush=True)
