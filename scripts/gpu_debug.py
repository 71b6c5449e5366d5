#!/usr/bin/env python3
"""Developer diagnostics (run on the GPU box): compare the HIP kernel's intermediates with the
oracle for env 0, stage by stage. Not a test; tests/test_gpu_parity.py is."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "trex-gym_amd"))
from oracle import oracle as O, trex_model as tm  # noqa: E402
from trex_gym import _capi  # noqa: E402

np.set_printoptions(precision=5, suppress=True, linewidth=200)
dev = torch.device("cuda:0")
om = tm.compile_model(O.default_asset_urdf())
order = om["obs_order"]
nb, nj = 26, 25


def rel(a, b):
    return np.abs(a - b).max() / max(1e-12, np.abs(b).max())


def compare_substep(name, state63, motors_on, target=None):
    """one substep from the same state on GPU (debug dump) and oracle"""
    orc = O.Oracle(om, params=dict(substeps=1))
    s = orc.new_state()
    orc.set_state(s, state63)
    orc.set_motors_on(s, motors_on)
    qdd_o, ba_o = orc.forward_dynamics(s, None, with_damping=True)
    minv_o = orc.minv(s)
    m = _capi.Model()
    m.set_param("substeps", 1)
    b = _capi.Batch(m, 4)
    st = torch.tensor(np.tile(state63, (4, 1)), dtype=torch.float32, device=dev)
    b.set_state(st)
    act = torch.tensor(np.tile(target if target is not None else np.zeros(nj), (4, 1)), dtype=torch.float32, device=dev)
    obs = torch.zeros(4, 75, device=dev)
    dbg = torch.zeros(4096, device=dev)
    if motors_on:
        b.debug_step(act, obs, dbg)
    else:
        # reset kernel path would overwrite the state; emulate motors off with zero max force
        m2 = _capi.Model(); m2.set_param("substeps", 1); m2.set_param("motor_max_force", 0.0)
        b = _capi.Batch(m2, 4); b.set_state(st); b.debug_step(act, obs, dbg)
    torch.cuda.synchronize()
    D = dbg.cpu().numpy()
    out = torch.zeros(4, 63, device=dev)
    b.get_state(out)
    gs = out.cpu().numpy()
    assert np.abs(gs - gs[0]).max() == 0, "envs with identical input differ"
    if motors_on:
        orc.substep(s, target if target is not None else np.zeros(nj))
    else:
        orc.substep(s)
    os_ = orc.get_state(s)
    print("==", name)
    print(" qdd   rel err %.2e" % rel(D[order], qdd_o), " base acc (spatial vs classical differ in lin part) ang err %.2e" % rel(D[32:35], ba_o[:3]))
    minv_g = np.zeros((31, 31))
    # GPU columns: D[160+32*(j-1)+lane]; lane l: joint l (1..25) or base dof nb+k
    for j in range(1, nb):
        col = D[160 + 32 * (j - 1): 160 + 32 * j]
        minv_g[6 + np.arange(nj), 6 + j - 1] = col[1:nb]
    print(" Minv joint block rel err %.2e" % rel(minv_g[6:, 6:], minv_o[6:, 6:]))
    nc = int(D[128])
    bo, lo, po, do = orc.contacts(s)
    print(" contacts gpu %d oracle %d" % (nc, len(bo)))
    if nc or len(bo):
        gb = D[960 + 16 * np.arange(nc)].astype(int)
        gx = np.stack([D[960 + 16 * c + 1: 960 + 16 * c + 4] for c in range(nc)]) if nc else np.zeros((0, 3))
        gl = np.stack([D[960 + 16 * c + 11: 960 + 16 * c + 14] for c in range(nc)]) if nc else np.zeros((0, 3))
        print("  gpu bodies", gb, "\n  orc bodies", bo)
        if nc == len(bo) and np.all(gb == bo):
            print("  contact x err %.2e  dist err %.2e  lambda rel err %.2e" % (
                np.abs(gx + state63[:3] - po).max(), np.abs(D[960 + 16 * np.arange(nc) + 4] - do).max(), rel(gl, lo)))
            print("  lambda_n gpu", gl[:, 0], "\n  lambda_n orc", lo[:, 0])
    print(" state after substep: max abs err pos %.2e quat %.2e v %.2e w %.2e q %.2e qd %.2e (qd scale %.2f)" % (
        np.abs(gs[0, :3] - os_[:3]).max(), np.abs(gs[0, 3:7] - os_[3:7]).max(), np.abs(gs[0, 7:10] - os_[7:10]).max(),
        np.abs(gs[0, 10:13] - os_[10:13]).max(), np.abs(gs[0, 13:38] - os_[13:38]).max(),
        np.abs(gs[0, 38:] - os_[38:]).max(), np.abs(os_[38:]).max()))
    ob = obs.cpu().numpy()[0]
    oo = orc.observe(s)
    print(" motor torque rel err %.2e (scale %.1f)" % (rel(ob[50:], oo[50:]), np.abs(oo[50:]).max()))
    return gs[0], os_


def main():
    rng = np.random.default_rng(0)
    # A: start pose in the air, motors off
    st = np.zeros(63); st[2] = 3; st[6] = 1; st[13:38] = om["q_start"][order]
    compare_substep("start pose, motors off", st, False)
    # B: random airborne state with velocities, motors on
    st = np.zeros(63); st[:3] = [0.3, -0.2, 6.0]
    q = rng.normal(size=4); st[3:7] = q / np.linalg.norm(q)
    st[7:13] = rng.normal(size=6)
    lo, hi = om["q_lower"][order], om["q_upper"][order]
    st[13:38] = rng.uniform(0.8 * lo, 0.8 * hi); st[38:] = rng.normal(size=25)
    compare_substep("random airborne, motors on", st, True, rng.uniform(lo, hi))
    # C: joints past limits
    st2 = st.copy(); st2[13:38] = lo - 0.03
    compare_substep("past lower limits", st2, True, rng.uniform(lo, hi))
    # D: settled crouch on the ground (contacts)
    orc = O.Oracle(om)
    s = orc.new_state(); orc.reset(s)
    q0 = om["q_start"][order]
    for i in range(60):
        orc.step(s, q0)
    compare_substep("landing, in contact", orc.get_state(s), True, q0)
    for i in range(200):
        orc.step(s, q0)
    compare_substep("at rest on the ground", orc.get_state(s), True, q0)

    # E: trajectories
    m = _capi.Model()
    N = 8
    b = _capi.Batch(m, N)
    obs = torch.zeros(N, 75, device=dev); rew = torch.zeros(N, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev)
    pen = torch.zeros(N, 3, device=dev)
    b.reset(obs)
    torch.cuda.synchronize()
    orc = O.Oracle(om); s = orc.new_state(); oo = orc.reset(s)
    print("reset obs err", np.abs(obs.cpu().numpy()[0] - oo).max())
    act = torch.tensor(np.tile(q0, (N, 1)), dtype=torch.float32, device=dev)
    for i in range(120):
        b.step(act, obs, rew, done, pen)
        oo, r, p = orc.step(s, q0)
        if i % 10 == 9:
            g = obs.cpu().numpy()[0]
            print("step %3d  q err %.2e qd err %.2e tau relerr %.2e  reward gpu %.4f orc %.4f" % (
                i + 1, np.abs(g[:25] - oo[:25]).max(), np.abs(g[25:50] - oo[25:50]).max(), rel(g[50:], oo[50:]), rew[0].item(), r))
    import time
    N = 4096
    b = _capi.Batch(m, N)
    obs = torch.zeros(N, 75, device=dev); rew = torch.zeros(N, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev)
    b.reset(obs)
    lo_t = torch.tensor(lo, dtype=torch.float32, device=dev); hi_t = torch.tensor(hi, dtype=torch.float32, device=dev)
    act = lo_t + (hi_t - lo_t) * torch.rand(N, 25, device=dev)
    for i in range(10):
        b.step(act, obs, rew, done)
    torch.cuda.synchronize()
    ms = b.time_steps(act, obs, rew, done, 20)
    print("4096 envs: %.3f ms/step -> %.0f env-steps/s; launch info %s" % (ms, N / ms * 1e3, b.launch_info()))
    print("finite:", torch.isfinite(obs).all().item(), "reward mean", rew.mean().item())


if __name__ == "__main__":
    main()
