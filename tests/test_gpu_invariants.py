"""Physics invariants checked ON THE GPU PATH itself (not only on the oracle), and a second, tiny
model through the same kernel: the HIP code is not specialised to the T-rex tree."""
import os

import numpy as np
import pytest
import torch

from conftest import ASSET_URDF

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

PENDULUM = """<robot name='p'>
  <link name='base'><inertial><origin xyz='0 0 0' rpy='0 0 0'/><mass value='1000'/><inertia ixx='500' iyy='500' izz='500'/></inertial></link>
  <link name='arm'><inertial><origin xyz='0 0 -0.5' rpy='0 0 0'/><mass value='1'/><inertia ixx='0.02' iyy='0.02' izz='0.001'/></inertial></link>
  <link name='fore'><inertial><origin xyz='0 0 -0.3' rpy='0 0 0'/><mass value='0.5'/><inertia ixx='0.01' iyy='0.01' izz='0.001'/></inertial></link>
  <joint name='shoulder' type='revolute'><parent link='base'/><child link='arm'/><origin xyz='0 0 0' rpy='0 0 0'/><axis xyz='0 1 0'/><limit lower='-3' upper='3'/></joint>
  <joint name='elbow' type='revolute'><parent link='arm'/><child link='fore'/><origin xyz='0 0 -1' rpy='0 0 0'/><axis xyz='0 1 0'/><limit lower='-3' upper='3'/></joint>
</robot>"""


def test_free_fall_and_momentum_on_gpu(model):
    """Motors off, damping off, high above the floor: the COM accelerates at -g and horizontal momentum
    is conserved - evaluated from the GPU state alone."""
    from trex_gym import _capi
    m = _capi.Model(ASSET_URDF)
    m.set_param("link_damping", 0.0)
    m.set_param("motor_max_force", 0.0)
    m.set_param("substeps", 50)
    b = _capi.Batch(m, 4)
    rng = np.random.default_rng(0)
    st = np.zeros((4, 63), np.float32)
    st[:, 2] = 60.0
    q = rng.normal(size=(4, 4)); st[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    st[:, 7:13] = 0.3 * rng.normal(size=(4, 6))
    lo, hi = model["q_lower"][model["obs_order"]], model["q_upper"][model["obs_order"]]
    st[:, 13:38] = rng.uniform(0.5 * lo, 0.5 * hi, (4, 25))
    st[:, 38:] = 0.3 * rng.normal(size=(4, 25))
    # joint damping is a URDF property (1.0 N m s): its torques are internal, momentum is unaffected
    dev_st = torch.tensor(st, device=DEV)
    b.set_state(dev_st)
    b.set_motors_enabled(True)

    def com_and_momentum(state_row):
        # FK on the host from the GPU state (numpy oracle model only for masses/geometry)
        from oracle import oracle as O
        orc = O.Oracle(model)
        s = orc.new_state()
        orc.set_state(s, state_row.astype(np.float64))
        e = orc.energy(s)
        pos, rot = orc.body_poses(s)
        mt = model["mass"].sum()
        com = sum(model["mass"][i] * (pos[i] + rot[i] @ model["com"][i]) for i in range(model["nb"])) / mt
        return com, e["momentum"][3:6], mt
    c0 = [com_and_momentum(r) for r in st]
    obs = torch.zeros(4, 75, device=DEV); rew = torch.zeros(4, device=DEV); done = torch.zeros(4, dtype=torch.uint8, device=DEV)
    b.step(torch.zeros(4, 25, device=DEV), obs, rew, done)     # 50 substeps = 0.1 s
    out = torch.zeros(4, 63, device=DEV)
    b.get_state(out)
    st1 = out.cpu().numpy()
    t = 50 * 0.002
    for k in range(4):
        com1, mom1, mt = com_and_momentum(st1[k])
        com0, mom0, _ = c0[k]
        v0 = mom0 / mt
        want = com0 + v0 * t + np.array([0, 0, -0.5 * 9.81 * t * (t + 0.002)])   # semi-implicit Euler sum
        np.testing.assert_allclose(com1, want, atol=2e-4)
        np.testing.assert_allclose(mom1 - mom0, [0, 0, -mt * 9.81 * t], atol=2e-4 * mt)


def test_double_pendulum_model_through_the_same_kernel(tmp_path):
    """A 3-body URDF (heavy base + two-link pendulum), no hulls: kernel vs oracle on a different tree."""
    from oracle import oracle as O, trex_model as tm
    from trex_gym import _capi
    u = tmp_path / "pend.urdf"
    u.write_text(PENDULUM)
    om = tm.compile_model(str(u))
    assert om["nb"] == 3
    orc = O.Oracle(om, params=dict(link_damping=0.0))
    m = _capi.Model(str(u))
    m.set_param("link_damping", 0.0)
    m.set_start_pose([0, 0, 5.0], [0, 0, 0])
    m.set_start_angle("shoulder", 0.7)
    om["q_start"][om["joint_names"].index("shoulder")] = 0.7
    om["base_start_pos"] = np.array([0, 0, 5.0])
    orc = O.Oracle(om, params=dict(link_damping=0.0))
    b = _capi.Batch(m, 3)
    obs = torch.zeros(3, 6, device=DEV); rew = torch.zeros(3, device=DEV); done = torch.zeros(3, dtype=torch.uint8, device=DEV)
    b.reset(obs)
    s = orc.new_state()
    o0 = orc.reset(s)
    np.testing.assert_allclose(obs.cpu().numpy()[0], o0, atol=1e-6)
    a = torch.tensor([[0.2, -0.4]] * 3, device=DEV)
    for t in range(40):
        b.step(a, obs, rew, done)
        o, r, _ = orc.step(s, np.array([0.2, -0.4]))
    g = obs.cpu().numpy()
    assert (g == g[0]).all()
    np.testing.assert_allclose(g[0, :2], o[:2], atol=2e-5)
    np.testing.assert_allclose(g[0, 2:4], o[2:4], atol=2e-4)
    np.testing.assert_allclose(g[0, 4:], o[4:], atol=2e-3 * (np.abs(o[4:]).max() + 1))
    # the position motor pulls the joints toward their targets
    assert abs(g[0, 0] - 0.2) < abs(0.7 - 0.2) and abs(g[0, 1] + 0.4) < 0.4


def test_link_transforms_match_reference_fk_and_oracle(model, oracle64):
    """Rollout export: world pose of all 133 URDF link frames.
    (a) start pose with the ROOT LINK frame at identity == the fixture computed by the reference's own
        Transform algebra (tests/golden/urdf_reference.json, scripts/make_golden.py);
    (b) random states == oracle body poses composed with the link-in-body transforms."""
    import json
    import os
    from oracle import trex_model as tm
    from trex_gym.vec_env import TrexVecEnv
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "urdf_reference.json")))
    joints, links = tm.parse_urdf(ASSET_URDF)
    root = [l for l in links if l["name"] == model["body_names"][0]][0]
    n = 5
    st = np.zeros((n, 63), np.float32)
    st[0, 0:3] = root["inertial_origin"].t
    st[0, 3:7] = tm.matrix_to_quat(root["inertial_origin"].R)
    st[0, 13:38] = model["q_start"][model["obs_order"]]
    rng = np.random.default_rng(3)
    lo, hi = model["q_lower"][model["obs_order"]], model["q_upper"][model["obs_order"]]
    for e in range(1, n):
        st[e, 0:3] = rng.normal(size=3) + [0, 0, 4]
        q = rng.normal(size=4); st[e, 3:7] = q / np.linalg.norm(q)
        st[e, 13:38] = rng.uniform(lo, hi)
    v = TrexVecEnv(n, urdf_path=ASSET_URDF, device=DEV)
    v.reset()
    v.set_state(torch.tensor(st))
    out = v.link_transforms().cpu().numpy()
    names = [nm for nm, _ in v.model.links()]
    assert out.shape == (n, 133, 7) and names == model["link_names"]
    for k, nm in enumerate(names):                       # (a) the reference's FK, every link
        g = gold["fk_start_pose"][nm]
        np.testing.assert_allclose(out[0, k, :3], g["xyz"], atol=5e-6, err_msg=nm)
        np.testing.assert_allclose(tm.quat_to_matrix(out[0, k, 3:]), tm.quat_to_matrix(g["quat_xyzw"]), atol=5e-6, err_msg=nm)
    for e in range(1, n):                                # (b) oracle composition
        s = oracle64.new_state()
        oracle64.set_state(s, st[e].astype(np.float64))
        pos, rot = oracle64.body_poses(s)
        for k in range(133):
            b = model["link_body"][k]
            R = rot[b] @ model["link_tf"][k][:9].reshape(3, 3)
            p = pos[b] + rot[b] @ model["link_tf"][k][9:]
            np.testing.assert_allclose(out[e, k, :3], p, atol=2e-5)
            np.testing.assert_allclose(tm.quat_to_matrix(out[e, k, 3:]), R, atol=2e-5)
            assert out[e, k, 6] >= 0 and abs(np.linalg.norm(out[e, k, 3:]) - 1) < 1e-5
    # (c) the <visual> meshes (trex_batch_visual_transforms): env 0 == the fixture computed by the reference's parser
    # and Transform algebra (fk[link] * shape.origin, scripts/make_golden.py), every env == link pose x <origin>
    vis = v.visual_transforms().cpu().numpy()
    table = v.model.visuals()
    assert vis.shape == (n, 252, 7) and len(gold["visuals"]) == 252
    for k, (g, (file, link, xyz, quat)) in enumerate(zip(gold["visuals"], table)):
        assert file == g["file"] and names[link] == g["link"]
        np.testing.assert_allclose(vis[0, k, :3], g["pose_start"]["xyz"], atol=5e-6, err_msg=file)
        np.testing.assert_allclose(tm.quat_to_matrix(vis[0, k, 3:]), tm.quat_to_matrix(g["pose_start"]["quat_xyzw"]), atol=5e-6, err_msg=file)
        for e in range(1, n):
            Rl = tm.quat_to_matrix(out[e, link, 3:])
            np.testing.assert_allclose(vis[e, k, :3], out[e, link, :3] + Rl @ xyz, atol=2e-5)
            np.testing.assert_allclose(tm.quat_to_matrix(vis[e, k, 3:]), Rl @ tm.quat_to_matrix(quat), atol=2e-5)


def test_step_is_graph_capturable_and_stream_ordered():
    """The launch path allocates nothing and never syncs, so a step can be captured in a HIP graph on
    a side stream and replayed; replay == eager, bitwise."""
    from trex_gym.vec_env import TrexVecEnv
    n = 512
    a = torch.zeros(n, 25, device=DEV)
    eager = TrexVecEnv(n, urdf_path=ASSET_URDF, device=DEV)
    graph = TrexVecEnv(n, urdf_path=ASSET_URDF, device=DEV)
    eager.reset_tensor(); graph.reset_tensor()
    torch.cuda.synchronize()
    graph.batch.step_rows(a, graph.rows)   # (first call validates the buffers; not captured)
    eager.batch.step_rows(a, eager.rows)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):            # captures on a side stream; _capi passes torch's current stream down
        graph.batch.step_rows(a, graph.rows)
    for t in range(12):
        a.copy_(0.3 * torch.sin(torch.arange(25, device=DEV) + t).expand(n, 25))
        g.replay()
        eager.batch.step_rows(a, eager.rows)
    torch.cuda.synchronize()
    assert (graph.rows == eager.rows).all() and (graph.penalties == eager.penalties).all()
    assert (graph.get_state() == eager.get_state()).all()


def test_launch_info_and_event_timing(capi=None):
    from trex_gym.vec_env import TrexVecEnv
    v = TrexVecEnv(1000, urdf_path=ASSET_URDF, device=DEV)
    v.reset_tensor()
    info = v.batch.launch_info()
    # an even batch of at most 4096 envs: the pair form, two envs = two waves per workgroup (one env per WAVE either way)
    assert info["grid"] == 500 and info["block"] == 128 and info["alg_bytes_per_env_step"] == 912
    assert 0 < info["lds_bytes"] <= 20 * 1024          # 8 two-env workgroups per CU (4 waves per SIMD) must fit the 160 KB of LDS
    odd = TrexVecEnv(999, urdf_path=ASSET_URDF, device=DEV).batch.launch_info()
    assert odd["grid"] == 999 and odd["block"] == 64 and 0 < odd["lds_bytes"] <= 10 * 1024      # the single-env form
    ms = v.batch.time_steps(torch.zeros(1000, 25, device=DEV), torch.zeros(1000, 75, device=DEV),
                            torch.zeros(1000, device=DEV), torch.zeros(1000, dtype=torch.uint8, device=DEV), 5)
    assert 0.05 < ms < 50


def test_primitive_collision_mode_matches_oracle(model):
    """capsule / sphere collision primitives through the same kernel; oracle with the numpy
    fit. One env-step from states along a landing, plus rest statistics (weight carried, contact count)."""
    from oracle import oracle as O, trex_model as tm
    from trex_gym.vec_env import TrexVecEnv
    om = tm.use_primitive_collision(model, 0.2, 3, 4)
    orc = O.Oracle(om)
    q0 = model["q_start"][model["obs_order"]]
    lo, hi = model["q_lower"][model["obs_order"]], model["q_upper"][model["obs_order"]]
    rng = np.random.default_rng(9)
    s = orc.new_state()
    orc.reset(s)
    states, acts = [], []
    for t in range(240):
        orc.step(s, np.clip(q0 + 0.1 * rng.normal(size=25), lo, hi))
        if t % 8 == 0:
            states.append(orc.get_state(s).astype(np.float32))
            acts.append(np.clip(q0 + 0.1 * rng.normal(size=25), lo, hi).astype(np.float32))
    states, acts = np.array(states), np.array(acts)
    v = TrexVecEnv(len(states), urdf_path=ASSET_URDF, device=DEV, collision="primitives")
    assert v.model.array("hull_radius").size == 148
    v.reset()
    v.set_state(torch.tensor(states))
    obs, rew, _, _ = v.step(acts)
    cnt = torch.zeros(len(states), dtype=torch.int32, device=DEV)
    v.batch.contact_stats(cnt, None)
    touched = 0
    for k in range(len(states)):
        s2 = orc.new_state()
        orc.set_state(s2, states[k].astype(np.float64))
        o, r, _ = orc.step(s2, acts[k].astype(np.float64))
        np.testing.assert_allclose(obs[k, :25], o[:25], atol=1e-4)
        np.testing.assert_allclose(obs[k, 25:50], o[25:50], atol=5e-3 * max(1.0, np.abs(o[25:50]).max()))
        np.testing.assert_allclose(obs[k, 50:], o[50:], atol=5e-3 * np.abs(o[50:]).max() + 1.0)
        assert abs(rew[k] - r) <= 2e-3 * abs(r) + 1e-3
        nco = len(orc.contacts(s2)[0])
        assert cnt[k].item() == nco
        touched += nco > 0
    assert touched > 10
    # holding the start pose it lands and the primitives carry the weight, on the GPU too
    vr = TrexVecEnv(2, urdf_path=ASSET_URDF, device=DEV, collision="primitives")
    vr.reset()
    a = np.tile(q0.astype(np.float32), (2, 1))
    for t in range(400):
        vr.step(a)
    imp = torch.zeros(2, device=DEV)
    vr.batch.contact_stats(None, imp)
    w = model["mass"].sum() * 9.81
    st = vr.get_state().cpu().numpy()[0]
    assert abs(imp[0].item() / 0.002 - w) < 0.05 * w and 0.5 < st[2] < 3.0 and np.abs(st[7:13]).max() < 0.2


def test_cpp_caller_of_the_c_abi():
    """trex_capi_example: a C++ program on the C-ABI with hipMalloc buffers (no Python, no torch)."""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "trex-gym_amd", "trex_gym", "trex_capi_example")
    # (-m gpu: the GPU box runs the snapshot's prebuilt binary - its absence is a failure of build(), not a skip)
    assert os.path.exists(exe), "example binary not built (make -C trex-gym_amd/csrc example; __graft_entry__.build() does it)"
    r = subprocess.run([exe, ASSET_URDF, "1024", "50"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "26 bodies, 25 joints (132 URDF joints), 2181 hull vertices" in r.stdout
    assert "bad buffers refused" in r.stdout     # host pointer / short buffer -> TREX_E_INVALID, no GPU fault
    rate = float(r.stdout.strip().splitlines()[-1].split("=")[-1].split()[0])
    assert rate > 1e5
