#!/usr/bin/env python3
"""Where does a substep spend its cycles? Runs the s_memtime-stamped DIAGNOSTIC build
(make -C trex-gym_amd/csrc stamps; TREX_LIB=...libtrex_hip_stamps.so) with 4096 envs in a
contact-rich state and prints workgroup 0's cycle shares per phase. Shares, not lengths, are
meaningful (the stamps fence the scheduler)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
os.environ.setdefault("TREX_LIB", os.path.join(ROOT, "trex-gym_amd", "trex_gym", "libtrex_hip_stamps.so"))
sys.path.insert(0, os.path.join(ROOT, "trex-gym_amd"))
from trex_gym import _capi, sharding  # noqa: E402

NAMES = ["FK + velocities", "inertia, bias", "ABA pass 2 (LDS)", "base 6x6 inverse", "ABA pass 3 + vel update",
         "factorisation A", "diag of M^-1", "joint rows", "contact generation", "contact chain walk (x2)",
         "row staging + B build (x2)", "(unused)", "PGS (60 sweeps, both envs)", "results + integrate"]


def main():
    dev = torch.device("cuda:0")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096   # 512 = one wave per CU: a wave alone on its SIMD
    m = _capi.Model()
    b = _capi.Batch(m, n)
    obs = torch.zeros(n, 75, device=dev); rew = torch.zeros(n, device=dev); done = torch.zeros(n, dtype=torch.uint8, device=dev)
    b.reset(obs)
    ids = torch.arange(n, device=dev)
    # hold the start pose: every env lands and rests on toes + tail with 14 contact points, so that
    # workgroup 0 (the stamped one) is a HEAVY wave - the kernel is as slow as its slowest wave
    hold = torch.tensor(m.array("q_start")[m.array("obs_order").astype(int)], dtype=torch.float32, device=dev).repeat(n, 1)
    for t in range(150):
        b.step(hold, obs, rew, done)
    dbg = torch.zeros(4096, device=dev)
    tot = np.zeros(14)
    joint = 0.0
    for t in range(10):
        dbg.zero_()   # the stamps accumulate over the two per-env solves of a substep
        b.debug_step(hold, obs, dbg)
        torch.cuda.synchronize()
        d = dbg.cpu().numpy()
        tot += d[3000:3000 + 80].reshape(5, 16)[:, :14].sum(0)
        joint += d[3000:3000 + 80].reshape(5, 16)[:, 14].sum()
        limw = d[3000:3000 + 80].reshape(5, 16)[:, 15]
    cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    b.contact_stats(cnt, None)
    print("contacts per env: mean %.1f max %d (env0 %d)" % (cnt.float().mean().item(), cnt.max().item(), cnt[0].item()))
    tot /= 10
    for name, c in zip(NAMES, tot):
        print("%-32s %9.0f cycles  %5.1f %%" % (name, c, 100 * c / tot.sum()))
    print("%-32s %9.0f cycles per env-step (5 substeps)" % ("total", tot.sum()))
    sub2 = d[3100:3140].reshape(5, 8)[:, :5]
    print("contact generation marks (cycles after the previous phase stamp; last step, substeps x [scan, large, K, fast path, end]):\n", sub2)
    print("inside PGS: joint rows (limits+motors) %.0f cycles = %.1f per motor row; contact rows %.0f cycles = %.1f per row (3 x %d); limit mask %s"
          % (joint / 10, joint / 10 / (300 * 25), tot[12] - joint / 10, (tot[12] - joint / 10) / (300 * 3 * cnt[0].item() + 1e-9), cnt[0].item(), limw))


if __name__ == "__main__":
    main()
