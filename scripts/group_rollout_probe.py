"""Probe: the PPO rollout (act -> env step -> observe per step) with the envs as G independent half-batches on G streams,
so that one group's policy / statistics launches run beside the other group's step launch.
    python scripts/group_rollout_probe.py [G] [envs] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "trex-gym_amd"))
from trex_gym import _capi  # noqa: E402
from trex_gym.ppo import MlpPolicy  # noqa: E402
from trex_gym.vec_env import TrexVecEnv  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T = int(sys.argv[3]) if len(sys.argv) > 3 else 320
dev = torch.device("cuda:0")
m = n // G
envs = [TrexVecEnv(num_envs=m, device="cuda:0") for _ in range(G)]
kerns = [_capi.Policy(m, 75, 25, 64, 0) for _ in range(G)]
pol = MlpPolicy(kerns[0].layout, kerns[0].param_count, dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(G)]
sh = [s.cuda_stream for s in streams]
noise = [torch.randn(T, m, 25, device=dev) for _ in range(G)]
acts = [torch.zeros(m, 25, device=dev) for _ in range(G)]
bufs = [dict(obs=torch.zeros(m, 75, device=dev), act=torch.zeros(m, 25, device=dev), logp=torch.zeros(m, device=dev),
             val=torch.zeros(m, device=dev), rew=torch.zeros(m, device=dev), done=torch.zeros(m, device=dev),
             scale=torch.zeros(1, device=dev)) for _ in range(G)]
for g in range(G):
    envs[g].reset_tensor()
    kerns[g].observe(envs[g].rows, False, 0.99)
torch.cuda.synchronize()


def run(t0, t1):
    for t in range(t0, t1):
        for g in range(G):
            e, k, b = envs[g], kerns[g], bufs[g]
            with torch.cuda.stream(streams[g]):
                k.act(pol.theta, e.rows, noise[g][t], acts[g], b["obs"], b["act"], b["logp"], b["val"], clip_obs=10.0)
                e.step_tensor(acts[g])
                k.observe(e.rows, True, 0.99, b["rew"], b["done"], b["scale"])


run(0, 64)
torch.cuda.synchronize()
t = time.perf_counter()
run(64, T)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print("rollout, G = %d groups of %d envs: %.3f M env-steps/s (%.4f ms per step of all %d envs)" % (G, m, n * (T - 64) / dt / 1e6, dt / (T - 64) * 1e3, n), flush=True)
