#!/usr/bin/env python3
"""K1m (an experiment that is NOT in the product kernel: apply profiles/tools/k1m_matrix_path.patch first), how often the motor block
of a sweep is done by the linear map: needs the stamped diagnostic build with the counters
(make -C trex-gym_amd/csrc variant XFLAGS='-DTREX_STAMPS=1 -DTREX_MX_COUNT=1' SUFFIX=_mxc; TREX_LIB=.../libtrex_hip_mxc.so)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "trex-gym_amd"))
from trex_gym import _capi, sharding  # noqa: E402

dev = torch.device("cuda:0"); n = 4096; pre = 400
m = _capi.Model(); b = _capi.Batch(m, n)
obs = torch.zeros(n, 75, device=dev); rew = torch.zeros(n, device=dev); done = torch.zeros(n, dtype=torch.uint8, device=dev)
ids = torch.arange(n, device=dev)
lo, hi = torch.tensor(m.lower, dtype=torch.float32, device=dev), torch.tensor(m.upper, dtype=torch.float32, device=dev)
phase = (ids * 1000) // n
b.reset(obs)
for t in range(pre):
    b.step(sharding.synthetic_actions(ids, t, lo, hi, seed=0, device=dev), obs, rew, done)
    mk = (phase == ((-(t + 1)) % 1000)).to(torch.uint8)
    if bool(mk.any()):
        b.reset(obs, mk)
dbg = torch.zeros(4096 + 16 * n, device=dev)
b.debug_step(sharding.synthetic_actions(ids, pre, lo, hi, seed=0, device=dev), obs, dbg)
torch.cuda.synchronize()
full = dbg.cpu().numpy()[4096:4096 + 16 * n].reshape(16, n)
builds, fast, sweeps = full[13], full[14], full[15]
on = sweeps > 0
print("envs on the matrix path in all 5 substeps: %.1f %%; in some: %.1f %%" % (100 * (sweeps == 300).mean(), 100 * on.mean()))
print("of their sweeps, done by the map: %.1f %%; builds per substep %.2f" % (100 * fast[on].sum() / sweeps[on].sum(), builds[on].sum() / (sweeps[on].sum() / 60)))
print("histogram of the share done by the map:", np.histogram(fast[on] / sweeps[on], bins=10, range=(0, 1))[0])
print("histogram of builds per env-step (0..15):", np.bincount(builds[on].astype(int), minlength=16)[:16])
