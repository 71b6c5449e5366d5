"""Digest of a fixed scenario's trajectory (bench.py's: 4096 envs, fresh uniform actions, seed 0): sha1 of the row
block [obs | reward | done] every 50 steps. Two builds of the step kernel whose arithmetic is meant to be bitwise the
same (a re-scheduled block, a changed launch shape) must print the same lines:
    TREX_LIB=<other build> python scripts/state_digest.py [steps] [envs]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "trex-gym_amd"))
from trex_gym.vec_env import TrexVecEnv  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
env = TrexVecEnv(num_envs=n, device="cuda:0")
env.reset_tensor()
lo = torch.as_tensor(env.action_space.low, device="cuda:0")
hi = torch.as_tensor(env.action_space.high, device="cuda:0")
g = torch.Generator(device="cuda:0")
g.manual_seed(0)
h = hashlib.sha1()
cnt = torch.zeros(n, dtype=torch.int32, device="cuda:0")
for t in range(steps):
    a = lo + (hi - lo) * torch.rand((n, 25), generator=g, device="cuda:0")
    obs, rew, done = env.step_tensor(a)[:3]
    if (t + 1) % 50 == 0:
        h.update(obs.cpu().numpy().tobytes()); h.update(rew.cpu().numpy().tobytes())
        env.batch.contact_stats(cnt, None)
        print(t + 1, h.hexdigest()[:16], "contacts mean %.3f" % float(cnt.float().mean()), flush=True)
