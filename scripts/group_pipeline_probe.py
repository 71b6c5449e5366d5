"""Probe: the 4096 envs of one GPU as G independent batches (groups) of 4096/G envs, each stepping on its own stream with
no barrier between the groups - the tail of one group's launch (its few heavy waves) overlaps the bulk of another's.
    python scripts/group_pipeline_probe.py [G] [steps] [envs]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "trex-gym_amd"))
from trex_gym.vec_env import TrexVecEnv  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
pre = 600
dev = torch.device("cuda:0")
envs = [TrexVecEnv(num_envs=n // G, device="cuda:0") for _ in range(G)]
streams = [torch.cuda.Stream(device=dev) for _ in range(G)]
lo = torch.as_tensor(envs[0].action_space.low, device=dev)
hi = torch.as_tensor(envs[0].action_space.high, device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(0)
T = pre + steps
acts = [lo + (hi - lo) * torch.rand((T, n // G, 25), generator=gen, device=dev) for _ in range(G)]
for e in envs:
    e.reset_tensor()
torch.cuda.synchronize()


offset = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0     # stagger the groups' first launches: group g waits g * offset ms
sh = [s_.cuda_stream for s_ in streams]


join = len(sys.argv) > 5 and sys.argv[5] == "join"      # a barrier between the groups after every step (what a library-internal split would have)
evs = [torch.cuda.Event() for _ in range(G)]


def run(t0, t1):
    for t in range(t0, t1):
        for g in range(G):
            e = envs[g]
            e.batch.step_rows(acts[g][t], e.rows, e._penalties, stream=sh[g], done=e.done)
        if join:
            for g in range(G):
                evs[g].record(streams[g])
            for g in range(G):
                for h in range(G):
                    if h != g:
                        streams[g].wait_event(evs[h])


if offset > 0:
    for g in range(1, G):
        with torch.cuda.stream(streams[g]):
            torch.cuda._sleep(int(g * offset * 2.4e6))


run(0, pre)
torch.cuda.synchronize()
t = time.perf_counter()
run(pre, T)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print(("join every step, " if join else "") + "offset %.2f ms: G = %d groups of %d envs: %.3f M env-steps/s (%.4f ms per step of all %d envs)" % (offset, G, n // G, n * steps / dt / 1e6, dt / steps * 1e3, n), flush=True)
