"""Step the same scenario with two builds of the library (TREX_LIB) and compare the rows step by step:
    python scripts/lib_compare.py <suffix_a> <suffix_b> [envs] [steps]
Each build runs in a child process (the library is chosen at import); prints the first steps at which the rows differ,
the largest difference and where (env, column)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, %(pkg)r)
from trex_gym.vec_env import TrexVecEnv
n, steps = %(n)d, %(steps)d
env = TrexVecEnv(num_envs=n, device="cuda:0", max_episode_steps=%(limit)d)
env.reset_tensor()
lo = torch.as_tensor(env.action_space.low, device="cuda:0"); hi = torch.as_tensor(env.action_space.high, device="cuda:0")
g = torch.Generator(device="cuda:0"); g.manual_seed(0)
out = [env.rows.clone().cpu().numpy()]
for t in range(steps):
    a = lo + (hi - lo) * torch.rand((n, 25), generator=g, device="cuda:0")
    env.step_tensor(a)
    out.append(env.rows.clone().cpu().numpy())
np.save(%(out)r, np.stack(out))
"""


def run(sfx, n, steps, limit, out):
    lib = os.path.join(ROOT, "trex-gym_amd", "trex_gym", "libtrex_hip%s.so" % sfx)
    env = dict(os.environ, TREX_LIB=lib)
    subprocess.check_call([sys.executable, "-c", CHILD % dict(pkg=os.path.join(ROOT, "trex-gym_amd"), n=n, steps=steps, limit=limit, out=out)], env=env)
    return np.load(out)


if __name__ == "__main__":
    a, b = sys.argv[1], sys.argv[2]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
    limit = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    A = run("" if a == "product" else a, n, steps, limit, "/tmp/cmp_a.npy")
    B = run("" if b == "product" else b, n, steps, limit, "/tmp/cmp_b.npy")
    same = True
    for t in range(A.shape[0]):
        d = np.abs(A[t] - B[t])
        if d.max() > 0 or not np.array_equal(np.isnan(A[t]), np.isnan(B[t])):
            same = False
            e, c = np.unravel_index(np.nanargmax(d), d.shape)
            print("step %d: %d of %d envs differ, max |diff| %.3e at env %d column %d (%.6g vs %.6g)" % (
                t, int((d.max(1) > 0).sum()), n, d.max(), e, c, A[t, e, c], B[t, e, c]))
            if t > 6 and d.max() > 1e-3:
                break
    print("BITWISE EQUAL" if same else "DIFFERENT")
