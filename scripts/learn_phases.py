#!/usr/bin/env python3
"""Where learn_grad_kernel's time goes: s_memtime at the barriers of every workgroup's first wave (DIAGNOSTIC build:
make -C trex-gym_amd/csrc variant XFLAGS=-DTREX_LEARN_STAMPS=1 SUFFIX=_lstamps; TREX_LIB selects it). One PPO update of the
bench workload (4096 envs, 32 steps, 32 minibatches of 4096); the stamps are those of the LAST minibatch step."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "trex-gym_amd"))
from trex_gym import _capi  # noqa: E402
from trex_gym.ppo import PPO  # noqa: E402
from trex_gym.trex_train import build_environment  # noqa: E402

NAMES = ["stage theta + obs tile (to the first barrier)", "forward layer 1 -> B1", "forward layer 2 -> B2", "output layer, loss -> B3",
         "delta 2 | grad W3 -> B4", "delta 1 | grad W2 -> B5", "grad W1, grad b1"]


def main():
    env = build_environment(4096)
    agent = PPO(env, nsteps=32, nminibatches=32, noptepochs=1, seed=0, use_graphs=False)
    for _ in range(3):
        b = agent.collect()
    agent.update(b)
    torch.cuda.synchronize()
    n = 256 * 12
    host = (C.c_ulonglong * n)()
    rc = _capi.lib.trex_policy_debug_learn_stamps(host, n)
    assert rc == 0, rc
    t = np.array(host, dtype=np.float64).reshape(256, 12)
    d = np.diff(t[:, :8], axis=1)
    print("256 workgroups (128 tiles x 2 nets); s_memtime ticks")
    print("workgroup duration (wave 0, first to last stamp): mean %.0f max %.0f" % ((t[:, 7] - t[:, 0]).mean(), (t[:, 7] - t[:, 0]).max()))
    for net, nm in ((0, "policy"), (1, "value")):
        sel = slice(128 * net, 128 * net + 128)
        tot = (t[sel, 7] - t[sel, 0]).mean()
        print("%s net: duration mean %.0f" % (nm, tot))
        for k, name in enumerate(NAMES):
            print("    %-48s %8.0f %5.1f %%" % (name, d[sel, k].mean(), 100 * d[sel, k].mean() / tot))
        print("    of the staging: up to the row-table barrier %.0f; of the output phase: up to the last MFMA issued %.0f" % (
            (t[sel, 9] - t[sel, 0]).mean(), (t[sel, 8] - t[sel, 3]).mean() if net == 0 else float("nan")))


if __name__ == "__main__":
    main()
