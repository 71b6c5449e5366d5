"""GPU parity tests proper (run with -m gpu on an MI355X): the HIP path, called through the C-ABI
(trex_gym._capi -> libtrex_hip.so), against the f64 CPU oracle on the same inputs, against the
committed golden rollouts, and - at BASELINE's full 4096 envs - through size-independent properties.

Stated tolerances (f32 kernel vs f64 oracle; PGS amplifies rounding in contact):
  one env-step from an identical state: |dq| <= 1e-4 rad, |dqd| <= 3e-3 * max(1, |qd|_inf),
  motor torque <= 3e-3 * max|tau| over the unsaturated joints (+1 N m), saturated joints saturated alike,
  reward <= 2e-3 relative (+1e-3, + a tenth of what the qd / tau tolerances allow in the energy term)
  contact-free trajectories (8..25 steps): |dq| <= 1e-4, |dqd| <= 5e-4 * max(1, |qd|_inf)
  K steps THROUGH contact (K = 5, 10): the kernel separates from the f64 oracle no faster than the oracle's own f32 build does
  (per state <= 3 x that build's separation + a floor; over the 50 landing states the median ratio is below 1)
Measured over 252 states, 152 of them in contact (scripts/parity_stats.py, profiles/r03_parity_stats.txt, unchanged by round 4's
kernel edits - bitwise the same rows): max |dq| 3.4e-5, |dqd| 1.43e-3, torque 1.61e-3, reward 5.5e-4 (medians 1e-7 .. 2e-6);
airborne states 2e-7 .. 7e-6. 1.5e-3 is the kernel's one-step bound on these states: 60 sweeps over up to 64 coupled rows in
f32 on ill-conditioned contact states, in a summation order that differs from the oracle's (exact-arithmetic ablation builds
show the v_rsq / v_rcp / series shortcuts are not the cause; the oracle's OWN f32 build is 2.1e-4 .. 3.2e-3 off its f64 build
on the landing states, scripts/kstep_separation.py). The rate / torque tolerance was 5e-3 until round 3; it is 3e-3 now: 2x the
measured bound.
"""
import os

import numpy as np
import pytest
import torch

from conftest import ASSET_URDF

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_rollout.npz"))
DEV = "cuda:0"


@pytest.fixture(scope="module")
def capi():
    from trex_gym import _capi
    return _capi


def make_vec(n, **kw):
    from trex_gym.vec_env import TrexVecEnv
    return TrexVecEnv(n, urdf_path=ASSET_URDF, device=DEV, **kw)


def assert_step_close(g_obs, o_obs, g_rew=None, o_rew=None, what="", q_atol=1e-4, loosen=1.0):
    J = 25
    assert np.isfinite(g_obs).all(), what
    np.testing.assert_allclose(g_obs[:J], o_obs[:J], atol=q_atol, rtol=0, err_msg=what + " q")
    np.testing.assert_allclose(g_obs[J:2 * J], o_obs[J:2 * J], atol=loosen * 3e-3 * max(1.0, np.abs(o_obs[J:2 * J]).max()),
                               rtol=0, err_msg=what + " qd")
    # motor torque: a joint saturated in the oracle (3e5 N m, trex_robot.py:260) must be saturated with the same sign;
    # the others are compared on the scale of the largest UNsaturated torque (a saturated neighbour must not hide
    # an error of hundreds of N m)
    gt, ot = g_obs[2 * J:], o_obs[2 * J:]
    sat = np.abs(ot) >= 0.999 * 3.0e5
    assert np.all(np.abs(gt[sat]) >= 0.999 * 3.0e5) and np.all(np.sign(gt[sat]) == np.sign(ot[sat])), what + " saturated tau"
    if (~sat).any():
        tscale = np.abs(ot[~sat]).max()
        np.testing.assert_allclose(gt[~sat], ot[~sat], atol=3e-3 * tscale + 1.0, rtol=0, err_msg=what + " tau")
    if g_rew is not None:
        # reward = -lift - drift - w_e sum|qd tau| (trex_env.py:186-192): 2e-3 relative on the whole, plus what the
        # stated qd / tau tolerances allow in the energy term (w_e = 0.005, the default of every test here)
        qd_tol = 3e-3 * max(1.0, np.abs(o_obs[J:2 * J]).max())
        tau_tol = np.where(sat, 1e-3 * 3.0e5, 3e-3 * (np.abs(ot[~sat]).max() if (~sat).any() else 0.0) + 1.0)
        energy_tol = 0.005 * np.sum(np.abs(o_obs[J:2 * J]) * tau_tol + np.abs(ot) * qd_tol)
        assert abs(g_rew - o_rew) <= 2e-3 * abs(o_rew) + 1e-3 + 0.1 * energy_tol, (what, g_rew, o_rew)


def test_native_library_is_the_one_loaded(capi):
    assert os.path.exists(capi.LIB_PATH)
    assert "libtrex_hip.so" in open("/proc/self/maps").read()


def test_reset_matches_oracle(oracle64):
    v = make_vec(5)
    obs = v.reset()
    s = oracle64.new_state()
    want = oracle64.reset(s)
    for e in range(5):
        np.testing.assert_allclose(obs[e], want, atol=2e-6)
    st = v.get_state().cpu().numpy()
    np.testing.assert_allclose(st[0], oracle64.get_state(s), atol=2e-6)
    np.testing.assert_allclose(v.head_position().cpu().numpy()[0], oracle64.head_position(s), atol=1e-5)


@pytest.mark.parametrize("name", ["zero", "crouch", "random"])
def test_golden_rollout_contact_free_window(name):
    """Committed oracle vectors, first 20 env-steps (before any contact in all three plans)."""
    v = make_vec(2)
    obs = v.reset()
    np.testing.assert_allclose(obs[0], GOLD[name + "_obs"][0], atol=2e-6)
    acts = GOLD[name + "_actions"]
    n = 8 if name != "crouch" else 25   # zero/random actions drive the legs into the ground early
    for t in range(n):
        a = np.tile(acts[t], (2, 1)).astype(np.float32)
        obs, rew, done, info = v.step(a)
        want = GOLD[name + "_obs"][t + 1]
        np.testing.assert_allclose(obs[0, :25], want[:25], atol=1e-4, err_msg="q step %d" % t)
        np.testing.assert_allclose(obs[0, 25:50], want[25:50], atol=5e-4 * max(1.0, np.abs(want[25:50]).max()),
                                   err_msg="qd step %d" % t)
        want_r = GOLD[name + "_reward"][t]
        assert abs(rew[0] - want_r) <= 1e-3 * abs(want_r) + 1e-4
        assert not done.any() and info[0] == {}


@pytest.mark.parametrize("name", ["zero", "crouch", "random"])
def test_one_step_parity_from_golden_states(name, oracle64):
    """Every state of the golden rollouts (free fall, impact, rest) -> ONE env-step on GPU and oracle."""
    states = GOLD[name + "_state"][:-1]
    acts = GOLD[name + "_actions"]
    n = len(acts)
    v = make_vec(n)
    v.reset()
    v.set_state(torch.tensor(states, dtype=torch.float32), motors_enabled=True)
    obs, rew, done, _ = v.step(acts.astype(np.float32))
    for t in range(n):
        s = oracle64.new_state()
        oracle64.set_state(s, states[t].astype(np.float32).astype(np.float64))
        o, r, p = oracle64.step(s, acts[t].astype(np.float32).astype(np.float64))
        assert_step_close(obs[t], o, rew[t], r, "%s state %d" % (name, t))


def test_one_step_parity_in_contact_and_at_rest(oracle64, model):
    """States sampled along a 300-step landing (contacts, friction, joint stops under load)."""
    q0 = model["q_start"][model["obs_order"]]
    lo, hi = model["q_lower"][model["obs_order"]], model["q_upper"][model["obs_order"]]
    rng = np.random.default_rng(5)
    s = oracle64.new_state()
    oracle64.reset(s)
    states, acts = [], []
    for t in range(300):
        a = np.clip(q0 + 0.15 * rng.normal(size=25), lo, hi)
        oracle64.step(s, a)
        if t % 6 == 0:
            states.append(oracle64.get_state(s).astype(np.float32))
            acts.append(np.clip(q0 + 0.15 * rng.normal(size=25), lo, hi).astype(np.float32))
    states, acts = np.array(states), np.array(acts)
    v = make_vec(len(states))
    v.reset()
    v.set_state(torch.tensor(states))
    obs, rew, _, _ = v.step(acts)
    cnt = torch.zeros(len(states), dtype=torch.int32, device=DEV)
    v.batch.contact_stats(cnt, None)
    n_contact_states = 0
    for t in range(len(states)):
        s2 = oracle64.new_state()
        oracle64.set_state(s2, states[t].astype(np.float64))
        o, r, p = oracle64.step(s2, acts[t].astype(np.float64))
        assert_step_close(obs[t], o, rew[t], r, "landing state %d" % t)
        nco = len(oracle64.contacts(s2)[0])
        assert cnt[t].item() == nco
        n_contact_states += nco > 0
    assert n_contact_states > 20


def test_long_rollout_statistics(oracle64, model):
    """Through contact trajectories decorrelate (chaos), so compare what the physics fixes:
    rest height, carried weight, contact count (SURVEY 7 'contact chaos')."""
    q0 = model["q_start"][model["obs_order"]].astype(np.float32)
    v = make_vec(4)
    v.reset()
    a = np.tile(q0, (4, 1))
    for t in range(400):
        obs, rew, _, _ = v.step(a)
    s = oracle64.new_state()
    oracle64.reset(s)
    for t in range(400):
        o, r, _ = oracle64.step(s, q0)
    st = v.get_state().cpu().numpy()[0]
    so = oracle64.get_state(s)
    assert abs(st[2] - so[2]) < 0.02
    assert np.abs(st[7:13]).max() < 0.1
    imp = torch.zeros(4, device=DEV)
    cnt = torch.zeros(4, dtype=torch.int32, device=DEV)
    v.batch.contact_stats(cnt, imp)
    w = model["mass"].sum() * 9.81
    assert abs(imp[0].item() / 0.002 - w) < 0.05 * w
    assert cnt[0].item() == len(oracle64.contacts(s)[0])
    assert abs(rew[0] - r) < 0.02 * abs(r) + 0.05


def test_joint_limit_rows(oracle64, oracle32, model):
    """Every joint 0.05 rad past its stop with saturated motors pushing further: 25 limit rows fight 25
    motor rows. 60 PGS sweeps are far from converged on the neck chain (947 kg cranium behind a 12 kg
    atlas, tests/test_oracle_physics.py::test_joint_limit_rows), so rounding is amplified there: a 1e-7
    relative perturbation of the state moves the f64 result by 1e-4 of the velocity scale, and the f32 build
    of the ORACLE (single common frame: differences of m r^2-sized terms) is 1.4 % off its f64 build on the
    atlas / cervical joints. The kernel's body-local frames keep it at 2.1e-3 (scripts/limit_err.py).
    Tolerance: 5e-3 of the velocity scale against the f64 oracle."""
    lo = model["q_lower"][model["obs_order"]]
    st = np.zeros((3, 63), np.float32)
    st[:, 2] = 50
    st[:, 6] = 1
    st[:, 13:38] = lo - 0.05
    v = make_vec(3)
    v.reset()
    v.set_state(torch.tensor(st))
    a = np.tile(lo, (3, 1)).astype(np.float32)
    obs, _, _, _ = v.step(a)
    s = oracle64.new_state()
    oracle64.set_state(s, st[0].astype(np.float64))
    o, _, _ = oracle64.step(s, a[0].astype(np.float64))
    scale = np.abs(o[25:50]).max()
    np.testing.assert_allclose(obs[0, :25], o[:25], atol=1e-4 + 0.01 * 5e-3 * scale)
    np.testing.assert_allclose(obs[0, 25:50], o[25:50], atol=5e-3 * scale)
    s32 = oracle32.new_state()
    oracle32.set_state(s32, st[0].astype(np.float64))
    o32, _, _ = oracle32.step(s32, a[0].astype(np.float64))
    # the kernel is closer to the f64 oracle than the oracle's own f32 build is
    assert np.abs(obs[0, 25:50] - o[25:50]).max() < np.abs(o32[25:50] - o[25:50]).max()
    assert np.all(obs[0, :25] > lo - 0.05)


def test_domain_randomisation_matches_oracle(oracle64, model):
    rng = np.random.default_rng(1)
    n = 6
    ms = rng.uniform(0.8, 1.2, (n, 26)).astype(np.float32)
    mu = rng.uniform(0.5, 1.25, n).astype(np.float32)
    states = GOLD["crouch_state"][30:30 + n].astype(np.float32)   # around touchdown
    acts = GOLD["crouch_actions"][:n].astype(np.float32)
    v = make_vec(n)
    v.reset()
    v.set_domain(torch.tensor(ms), torch.tensor(mu))
    v.set_state(torch.tensor(states))
    obs, rew, _, _ = v.step(acts)
    differs = 0
    for e in range(n):
        s = oracle64.new_state()
        oracle64.set_domain(s, ms[e].astype(np.float64), float(mu[e]))
        oracle64.set_state(s, states[e].astype(np.float64))
        o, r, _ = oracle64.step(s, acts[e].astype(np.float64))
        assert_step_close(obs[e], o, rew[e], r, "domain env %d" % e)
        s0 = oracle64.new_state()
        oracle64.set_state(s0, states[e].astype(np.float64))
        o0, _, _ = oracle64.step(s0, acts[e].astype(np.float64))
        differs += np.abs(o0 - o).max() > 1e-3
    assert differs >= n - 1   # the randomisation really changes the dynamics


def test_action_repeat_and_params(oracle64, model):
    from oracle import oracle as O
    orc = O.Oracle(model, params=dict(substeps=10, iterations=20))
    v = make_vec(2, action_repeat=2, params=dict(iterations=20))
    obs = v.reset()
    s = orc.new_state()
    orc.reset(s)
    a = GOLD["random_actions"][0].astype(np.float32)
    obs, rew, _, _ = v.step(np.tile(a, (2, 1)))
    o, r, _ = orc.step(s, a.astype(np.float64))
    assert_step_close(obs[0], o, rew[0], r, "action_repeat=2")


# ---------------------------------------------------------------- full-size properties (4096 envs)
N_FULL = 4096


@pytest.fixture(scope="module")
def full(model):
    v = make_vec(N_FULL)
    lo = torch.tensor(model["q_lower"][model["obs_order"]], dtype=torch.float32, device=DEV)
    hi = torch.tensor(model["q_upper"][model["obs_order"]], dtype=torch.float32, device=DEV)
    return v, lo, hi


def test_full_identical_envs_stay_bitwise_identical(full):
    v, lo, hi = full
    v.reset_tensor()
    g = torch.Generator(device=DEV).manual_seed(0)
    for t in range(40):    # through first contact
        a = (lo + (hi - lo) * torch.rand(1, 25, device=DEV, generator=g)).expand(N_FULL, 25).contiguous()
        obs, rew, done = v.step_tensor(a)
    assert torch.isfinite(obs).all()
    assert (obs == obs[0:1]).all() and (rew == rew[0]).all()
    st = v.get_state()
    assert (st == st[0:1]).all()


def test_full_envs_are_independent_and_deterministic(full):
    """Permuting the action rows permutes the results; repeating the run reproduces them bitwise."""
    v, lo, hi = full
    g = torch.Generator(device=DEV).manual_seed(1)
    acts = lo + (hi - lo) * torch.rand(30, N_FULL, 25, device=DEV, generator=g)
    perm = torch.randperm(N_FULL, device=DEV, generator=g)

    def run(a_all):
        v.reset_tensor()
        for t in range(a_all.shape[0]):
            obs, rew, _ = v.step_tensor(a_all[t].contiguous())
        return obs.clone(), rew.clone(), v.get_state()
    o1, r1, s1 = run(acts)
    o2, r2, s2 = run(acts)
    assert (o1 == o2).all() and (r1 == r2).all() and (s1 == s2).all()
    o3, r3, s3 = run(acts[:, perm])
    assert (o3 == o1[perm]).all() and (r3 == r1[perm]).all() and (s3 == s1[perm]).all()
    assert torch.isfinite(o1).all()
    assert (o1[:, 50:].abs() <= 3.0e5 * (1 + 1e-6)).all()      # motor clamp, trex_robot.py:260
    assert o1.std(0).max() > 0                                 # envs really differ


def test_wave_balance_and_contact_budget(full):
    """The launch ranks envs by their previous contact count and deals the ranks over the SIMDs (balance
    kernel); an env's results must not depend on its history-dependent slot. And the contact budget: never
    more than 13 points."""
    v, lo, hi = full
    g = torch.Generator(device=DEV).manual_seed(7)
    v.reset_tensor()
    for t in range(120):   # land: contact counts from 0 to the budget
        obs, rew, _ = v.step_tensor((lo + (hi - lo) * torch.rand(N_FULL, 25, device=DEV, generator=g)).contiguous())
    cnt = torch.zeros(N_FULL, dtype=torch.int32, device=DEV)
    v.batch.contact_stats(cnt, None)
    assert int(cnt.max()) <= 13 and int(cnt.max()) >= 8 and int(cnt.min()) == 0
    st = v.get_state()
    a = (lo + (hi - lo) * torch.rand(N_FULL, 25, device=DEV, generator=g)).contiguous()
    o1, r1, _ = v.step_tensor(a)
    o1, r1, s1 = o1.clone(), r1.clone(), v.get_state()
    # same states, but the stored contact counts (hence the slot of every env) come from a different history
    v.reset_tensor()
    v.step_tensor(a)
    v.set_state(st)
    o2, r2, _ = v.step_tensor(a)
    assert (o1 == o2).all() and (r1 == r2).all() and (s1 == v.get_state()).all()


@pytest.mark.parametrize("n", [2500, 5000])
def test_rank_lists_cover_every_env_at_ragged_sizes(n, model):
    """The env-to-wave assignment is made inside the step kernel from lists the previous launch filed (two list sets,
    flipped by the last wave to end; blocks of 1024 ranks, the last one partial here): over many launches, through
    landing, EVERY env must be stepped exactly once per launch - the trajectories equal those of launches that keep
    env k in workgroup k (trex_batch_set_wave_balance(0)), bitwise."""
    lo = torch.tensor(model["q_lower"][model["obs_order"]], dtype=torch.float32, device=DEV)
    hi = torch.tensor(model["q_upper"][model["obs_order"]], dtype=torch.float32, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(5)
    acts = [(lo + (hi - lo) * torch.rand(n, 25, device=DEV, generator=g)).contiguous() for _ in range(70)]

    def run(no_balance):
        v = make_vec(n, max_episode_steps=50)
        if no_balance:
            v.batch.set_wave_balance(0)
        v.reset_tensor()
        v.set_episode_steps(torch.arange(n, device=DEV, dtype=torch.int32) % 50)   # resets inside the launches, too
        for a in acts:
            o, r, d = v.step_tensor(a)
        cnt = torch.zeros(n, dtype=torch.int32, device=DEV)
        v.batch.contact_stats(cnt, None)
        return o.clone(), r.clone(), v.get_state().clone(), cnt

    o1, r1, s1, c1 = run(False)
    o2, r2, s2, c2 = run(True)
    assert int(c1.max()) >= 6 and int(c1.min()) == 0          # the counts that drive the ranking did spread
    assert (o1 == o2).all() and (r1 == r2).all() and (s1 == s2).all() and (c1 == c2).all()


def test_bad_tensors_are_refused(full, capi):
    """Host memory, a wrong dtype, a short or non-contiguous buffer would be misread or fault the GPU:
    the binding refuses them with TREX_E_INVALID (and the C-ABI checks raw pointers again:
    tests/test_gpu_invariants.py::test_cpp_caller_of_the_c_abi)."""
    v, lo, hi = full
    obs, rew = torch.zeros(N_FULL, 75, device=DEV), torch.zeros(N_FULL, device=DEV)
    done, a = torch.zeros(N_FULL, dtype=torch.uint8, device=DEV), torch.zeros(N_FULL, 25, device=DEV)
    for bad in ((a.cpu(), obs, rew, done),                    # host memory
                (a, obs, rew, done.to(torch.int32)),          # wrong dtype
                (a, obs[: N_FULL // 2], rew, done),           # too short
                (a, v.obs, rew, done),                        # a strided view (columns of the row block)
                (a.double(), obs, rew, done)):
        with pytest.raises(capi.TrexError) as ei:
            v.batch.step(*bad)
        assert ei.value.code == capi.E_INVALID
    with pytest.raises(capi.TrexError):
        v.batch.step_rows(a, torch.zeros(N_FULL, 76, device=DEV))   # row block narrower than 3J + 2
    v.batch.step(a, obs, rew, done)                           # and the well-formed call still works


def test_validated_allocations_are_cached_by_range_and_can_be_forgotten():
    """The C-ABI validates every caller ALLOCATION once: any pointer into a validated allocation with enough room behind
    it passes (a [T, n, 25] action pool presents a new pointer every step), a pointer with too little room behind it is
    refused, and trex_batch_forget_buffers() drops the cache (for callers that free buffers they have passed)."""
    n = 64
    v = make_vec(n)
    v.reset_tensor()
    pool = torch.zeros(5, n, 25, device=DEV)
    for t in range(5):
        v.step_tensor(pool[t])                              # five different pointers, one allocation
    tail = pool.reshape(-1)[-(n * 25 - 8):]                 # starts 8 floats late: too short for [n, 25]
    with pytest.raises(Exception):
        v.batch.step_rows(tail, v.rows, None, done=v.done)
    v.batch.forget_buffers()
    o1, r1, _ = v.step_tensor(pool[0])                      # validated afresh
    assert torch.isfinite(o1).all() and torch.isfinite(r1).all()


def test_full_clipping_equals_clipped_actions(full):
    v, lo, hi = full
    g = torch.Generator(device=DEV).manual_seed(2)
    raw = 4.0 * torch.randn(N_FULL, 25, device=DEV, generator=g)
    v.reset_tensor()
    o1, r1, _ = v.step_tensor(raw)
    o1, r1 = o1.clone(), r1.clone()
    v.reset_tensor()
    o2, r2, _ = v.step_tensor(torch.minimum(torch.maximum(raw, lo), hi))
    assert (o1 == o2).all() and (r1 == r2).all()


def test_two_row_blocks_written_in_turn():
    """row_buffers=2 (what the in-place pipelined gather of bench.py --gpus N relies on): successive steps write two
    row blocks in turn, the block of the step before last is left alone, results equal the one-block env's."""
    n = 256
    one, two = make_vec(n, max_episode_steps=7), make_vec(n, max_episode_steps=7, row_buffers=2)
    g = torch.Generator(device=DEV).manual_seed(3)
    one.reset_tensor(); two.reset_tensor()
    blocks = []
    for t in range(9):      # past the episode limit: the done column is exercised
        a = (0.3 * torch.randn(n, 25, device=DEV, generator=g)).contiguous()
        o1, r1, d1 = one.step_tensor(a)
        o2, r2, d2 = two.step_tensor(a)
        assert (o1 == o2).all() and (r1 == r2).all() and (d1 == d2).all()
        assert (two.done_f != 0).equal(d2)
        if t >= 1:
            assert two.rows.data_ptr() != blocks[-1][0]            # the other block ...
            assert torch.equal(blocks[-1][1], blocks[-1][2])       # ... and the previous one was not touched
        if t >= 2:
            assert two.rows.data_ptr() == blocks[-2][0]
        blocks.append((two.rows.data_ptr(), two.rows, two.rows.clone()))
    assert bool(d2.any()) is False and int(two.episode_steps.max()) <= 7


def test_reset_with_the_done_flags_of_the_last_step():
    """reset_tensor(done): the bool flags step_tensor hands out are a valid reset mask (same bytes as uint8), and the
    rows of the envs that were reset read reward 0, done 0 afterwards - a consumer that gathers `rows` right after a
    reset must not see the finished episode's flags."""
    a = torch.zeros(6, 25, device=DEV)
    v = make_vec(6, max_episode_steps=4)
    first = v.reset_tensor().clone()
    v.set_episode_steps(torch.tensor([3, 0, 3, 0, 0, 0], dtype=torch.int32))
    o, r, d = v.step_tensor(a)
    assert d.dtype == torch.bool and d.tolist() == [True, False, True, False, False, False]
    assert v.done_f.tolist() == [1.0, 0.0, 1.0, 0.0, 0.0, 0.0] and bool((v.rew != 0).all())
    keep_rew = v.rew.clone()
    obs = v.reset_tensor(d)                      # bool mask; d IS v.done
    assert (obs[0] == first[0]).all() and (obs[2] == first[2]).all()
    assert v.done.tolist() == [False] * 6 and v.done_f.tolist() == [0.0] * 6
    assert v.rew[0].item() == 0.0 and v.rew[2].item() == 0.0
    assert (v.rew[[1, 3, 4, 5]] == keep_rew[[1, 3, 4, 5]]).all()       # untouched rows keep their columns
    with pytest.raises(Exception):
        v.reset_tensor(torch.zeros(6, dtype=torch.int32, device=DEV))  # neither uint8 nor bool


def test_row_block_of_77_and_of_80_columns():
    """trex_batch_step_rows: a [n, 3J+2] row block (obs | reward | done) with the penalties in an array of their own ==
    the [n, 3J+5] block that carries them behind done (what TrexVecEnv allocates: whole 32-byte sectors per env)."""
    n = 50
    g = torch.Generator(device=DEV).manual_seed(31)
    a = (torch.rand(n, 25, device=DEV, generator=g) - 0.5).contiguous()
    wide, narrow = make_vec(n, penalties_in_rows=True), make_vec(n)
    wide.reset_tensor(); narrow.reset_tensor()
    rows77, pen = torch.zeros(n, 77, device=DEV), torch.zeros(n, 3, device=DEV)
    for _ in range(35):
        wide.step_tensor(a)
        narrow.batch.step_rows(a, rows77, pen)
    assert wide.rows.shape == (n, 80)
    assert torch.equal(wide.rows[:, :77], rows77) and torch.equal(wide.penalties, pen)
    assert float(pen.abs().sum()) > 0
    mask = torch.zeros(n, dtype=torch.uint8, device=DEV); mask[3] = 1
    wide.reset_tensor(mask)                                    # a reset zeroes the row's reward / done / penalty columns
    assert float(wide.rows[3, 75:80].abs().sum()) == 0.0 and float(wide.rows[4, 77:80].abs().sum()) > 0.0


def test_padded_rows_keep_the_callers_columns():
    """rows padded for alignment or embedded in a wider tensor (row_stride 96) have columns of the caller's own
    beyond 3J + 2 = 77. Without trex_batch_set_penalties_in_rows NOTHING beyond column 76 is written - by a step, a reset, a
    masked reset or step_many; with it, exactly the three penalty columns 77..79 are."""
    n, S = 40, 3
    g = torch.Generator(device=DEV).manual_seed(5)
    a = (torch.rand(n, 25, device=DEV, generator=g) - 0.5).contiguous()
    many = (torch.rand(S, n, 25, device=DEV, generator=g) - 0.5).contiguous()
    v = make_vec(n)
    ref = make_vec(n)
    rows = torch.full((n, 96), -7.0, device=DEV)
    rows_many = torch.full((S, n, 96), -7.0, device=DEV)
    v.batch.reset_rows(rows); ref.reset_tensor()
    assert torch.equal(rows[:, :77], ref.rows) and bool((rows[:, 77:] == -7.0).all())
    for _ in range(30):
        v.batch.step_rows(a, rows, None); ref.step_tensor(a)
    assert torch.equal(rows[:, :77], ref.rows) and bool((rows[:, 77:] == -7.0).all())
    mask = torch.zeros(n, dtype=torch.uint8, device=DEV); mask[::3] = 1
    v.batch.reset_rows(rows, mask)
    assert bool((rows[:, 77:] == -7.0).all())
    v.batch.step_many(many, rows_many)
    assert bool((rows_many[:, :, 77:] == -7.0).all()) and bool(torch.isfinite(rows_many[:, :, :77]).all())
    v.batch.set_penalties_in_rows(True)
    pen = torch.zeros(n, 3, device=DEV)
    v.batch.step_rows(a, rows, pen)
    assert torch.equal(rows[:, 77:80], pen) and float(pen.abs().sum()) > 0 and bool((rows[:, 80:] == -7.0).all())
    with pytest.raises(Exception):
        v.batch.step_rows(a, torch.zeros(n, 78, device=DEV), None)        # the wide form needs row_stride >= 3J + 5


def test_step_many_hands_out_the_penalties_of_every_step():
    """step_many_tensor used to leave `penalties` at the values from before the call."""
    n, S = 16, 4
    g = torch.Generator(device=DEV).manual_seed(6)
    acts = (torch.rand(S, n, 25, device=DEV, generator=g) - 0.5).contiguous()
    one, many = make_vec(n), make_vec(n)
    one.reset_tensor(); many.reset_tensor()
    want = []
    for s_ in range(S):
        one.step_tensor(acts[s_])
        want.append(one.penalties.clone())
    many.step_many_tensor(acts)
    assert torch.equal(many.penalties_many, torch.stack(want)) and torch.equal(many.penalties, want[-1])


def test_wave_balance_setting_is_a_batch_property():
    """trex_batch_set_wave_balance: -1 auto (on from 2048 envs), 0 off, 1 on. Results are bitwise independent of it -
    also when it is forced on below the automatic threshold and when it is switched while the batch is running."""
    n = 300
    g = torch.Generator(device=DEV).manual_seed(9)
    lo = torch.tensor(make_vec(1).model.lower, dtype=torch.float32, device=DEV)
    hi = torch.tensor(make_vec(1).model.upper, dtype=torch.float32, device=DEV)
    acts = [(lo + (hi - lo) * torch.rand(n, 25, device=DEV, generator=g)).contiguous() for _ in range(45)]
    outs = []
    for mode in (-1, 1, "switch"):
        v = make_vec(n)
        if mode == 1:
            v.batch.set_wave_balance(1)
        v.reset_tensor()
        for t, a in enumerate(acts):
            if mode == "switch":
                v.batch.set_wave_balance(1 if (t // 7) % 2 else 0)
            o, r, _ = v.step_tensor(a)
        outs.append((o.clone(), r.clone(), v.get_state().clone()))
    for o, r, s in outs[1:]:
        assert (o == outs[0][0]).all() and (r == outs[0][1]).all() and (s == outs[0][2]).all()
    with pytest.raises(Exception):
        v.batch.set_wave_balance(2)


def test_step_many_is_bitwise_the_same_steps_one_by_one(model):
    """trex_batch_step_many: S env-steps per launch (open-loop action sequences) == S calls of trex_batch_step_rows,
    bitwise - observations, rewards, done flags, penalties of EVERY step and the final state - with the episode limit
    ending episodes inside the launch at different steps (staggered ages), per-env domain randomisation, a non-finite
    env contained in the middle, and the wave balance on (2 500 envs, ragged last block)."""
    n, S = 2500, 12
    lo = torch.tensor(model["q_lower"][model["obs_order"]], dtype=torch.float32, device=DEV)
    hi = torch.tensor(model["q_upper"][model["obs_order"]], dtype=torch.float32, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(21)
    acts = (lo + (hi - lo) * torch.rand(S, n, 25, device=DEV, generator=g)).contiguous()
    pre = [(lo + (hi - lo) * torch.rand(n, 25, device=DEV, generator=g)).contiguous() for _ in range(40)]
    ms = 0.8 + 0.4 * torch.rand(n, 26, device=DEV, generator=g)
    fr = 0.5 + 0.75 * torch.rand(n, device=DEV, generator=g)

    def prepare():
        v = make_vec(n, max_episode_steps=9)
        v.set_domain(ms, fr)
        v.reset_tensor()
        for a in pre:                        # through touchdown: contacts, joint stops
            v.step_tensor(a)
        v.set_episode_steps(torch.arange(n, device=DEV, dtype=torch.int32) % 9)
        st = v.get_state()
        st[17, 20] = float("nan")            # env 17 will be contained in the first step
        v.set_state(st)
        return v

    a = prepare()
    rows_a, pen_a, done_a = [], [], []
    for s_ in range(S):
        a.step_tensor(acts[s_])
        rows_a.append(a.rows.clone()); pen_a.append(a.penalties.clone()); done_a.append(a.done.clone())
    b = prepare()
    b.batch.set_penalties_in_rows(True)                 # (explicit since round 4: a wide stride alone writes nothing beyond column 76)
    rows_b = torch.empty(S, n, 80, device=DEV)          # obs | reward | done | 3 penalties (the wide form of the row block)
    pen_b = torch.empty(S, n, 3, device=DEV)
    done_b = torch.zeros(S, n, dtype=torch.bool, device=DEV)
    b.batch.step_many(acts, rows_b, pen_b, done_b)
    b.batch.set_penalties_in_rows(False)                # (b's own row blocks have 77 columns)
    assert bool(done_b[0, 17]) and float(rows_b[0, 17, 75]) == 0.0       # containment inside the launch
    assert int(done_b.sum()) > S * n // 9 - n                              # episodes ended in every step
    for s_ in range(S):
        assert torch.equal(rows_b[s_, :, :77], rows_a[s_]), s_
        assert torch.equal(pen_b[s_], pen_a[s_]) and torch.equal(done_b[s_], done_a[s_]), s_
        assert torch.equal(rows_b[s_, :, 77:80], pen_b[s_])             # the penalties ride in the row block, too
    assert torch.equal(b.get_state(), a.get_state())
    assert torch.equal(b.episode_steps, a.episode_steps)
    ca, cb = torch.zeros(n, dtype=torch.int32, device=DEV), torch.zeros(n, dtype=torch.int32, device=DEV)
    a.batch.contact_stats(ca, None); b.batch.contact_stats(cb, None)
    assert torch.equal(ca, cb) and int(ca.max()) >= 6
    # the next ordinary step continues from the same state (balance lists included)
    nxt = (lo + (hi - lo) * torch.rand(n, 25, device=DEV, generator=g)).contiguous()
    oa, ra, _ = a.step_tensor(nxt)
    ob, rb, _ = b.step_tensor(nxt)
    assert torch.equal(oa, ob) and torch.equal(ra, rb)
    # the convenience wrapper
    rows_c = prepare().step_many_tensor(acts)
    assert torch.equal(rows_c, rows_b[:, :, :77])
    with pytest.raises(Exception):
        b.batch.step_many(acts, rows_b[:, :100])                           # rows of the wrong shape are refused


def test_full_masked_reset_and_state_round_trip(full):
    v, lo, hi = full
    g = torch.Generator(device=DEV).manual_seed(3)
    v.reset_tensor()
    for t in range(5):
        v.step_tensor(lo + (hi - lo) * torch.rand(N_FULL, 25, device=DEV, generator=g))
    before = v.get_state()
    obs_before = v.obs.clone()
    mask = (torch.rand(N_FULL, device=DEV, generator=g) < 0.3).to(torch.uint8)
    obs = v.reset_tensor(mask).clone()
    after = v.get_state()
    keep = mask == 0
    assert (after[keep] == before[keep]).all() and (obs[keep] == obs_before[keep]).all()
    fresh = make_vec(2)
    f_obs = torch.tensor(fresh.reset(), device=DEV)
    assert (obs[~keep] == f_obs[0]).all()
    assert (after[~keep] == fresh.get_state()[0]).all()
    # set_state(get_state()) is the identity
    v.set_state(after)
    assert (v.get_state() == after).all()


def test_reward_is_the_reference_formula(full):
    """r = -w_d (2.5 - z)^2 - w_k (x^2 + y^2) - w_e sum|qd tau| with the head COM (trex_env.py:186-192)."""
    v, lo, hi = full
    g = torch.Generator(device=DEV).manual_seed(4)
    v.reset_tensor()
    for t in range(35):
        obs, rew, _ = v.step_tensor(lo + (hi - lo) * torch.rand(N_FULL, 25, device=DEV, generator=g))
    head = v.head_position().double()
    power = (obs[:, 25:50].double() * obs[:, 50:].double()).abs().sum(1)
    want = -1.0 * (2.5 - head[:, 2]) ** 2 - 0.002 * (head[:, 0] ** 2 + head[:, 1] ** 2) - 0.005 * power
    assert torch.allclose(rew.double(), want, rtol=2e-4, atol=1e-3)
    assert torch.allclose(v.penalties.double().sum(1), -rew.double(), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("params", [dict(contact_margin=0.5), dict(max_contacts=4)])
def test_more_touching_bodies_than_contact_rows(model, params):
    """17 bodies inside an inflated margin against the 13-point budget, and 7 touching bodies against a budget of 4:
    the rows go to the deepest bodies, in the kernel as in the oracle. One-step parity along the landing, contact
    counts equal, and after the landing no hull vertex more than 2 cm below the floor."""
    from oracle import oracle as O
    orc = O.Oracle(model, params=params)
    q0 = model["q_start"][model["obs_order"]].astype(np.float32)
    s = orc.new_state()
    orc.reset(s)
    states = []
    for t in range(150):
        orc.step(s, q0.astype(np.float64))
        if t % 5 == 4:
            states.append(orc.get_state(s).astype(np.float32))
    states = np.array(states)
    v = make_vec(len(states), params=params)
    v.reset()
    v.set_state(torch.tensor(states))
    acts = np.tile(q0, (len(states), 1))
    obs, rew, _, _ = v.step(acts)
    cnt = torch.zeros(len(states), dtype=torch.int32, device=DEV)
    v.batch.contact_stats(cnt, None)
    budget = int(params.get("max_contacts", 13))
    over = 0
    for t in range(len(states)):
        s2 = orc.new_state()
        orc.set_state(s2, states[t].astype(np.float64))
        o, r, _ = orc.step(s2, q0.astype(np.float64))
        # (4 rows for 7 touching bodies is not a consistent support - the supporting set rotates, the ground force
        # jitters around 1.2 x the weight, tests/test_oracle_physics.py - and amplifies rounding: 1e-2 of the rate scale there,
        # as before round 4 tightened the general tolerance to 3e-3; measured 7.2e-3 / 2.4)
        assert_step_close(obs[t], o, rew[t], r, "budget state %d" % t, loosen=10.0 / 3.0 if budget < 7 else 1.0)
        assert cnt[t].item() == len(orc.contacts(s2)[0]) <= budget
        over += cnt[t].item() == budget
    assert over > 5                                   # the budget really was exhausted along the way
    for _ in range(50):
        v.step(acts)
    st = v.get_state().cpu().numpy().astype(np.float64)
    hs, hv = model["hull_start"], model["hull_xyz"]
    for e in (0, len(states) // 2, len(states) - 1):
        s3 = orc.new_state()
        orc.set_state(s3, st[e])
        pos, rot = orc.body_poses(s3)
        low = min((pos[b][2] + (rot[b].reshape(3, 3) @ hv[hs[b]:hs[b + 1]].T)[2]).min()
                  for b in range(model["nb"]) if hs[b + 1] > hs[b])
        assert low > 0.0005 - 0.02


def test_config4_size_on_one_gpu(oracle64, oracle32, model):
    """BASELINE config 4's batch (32 768 envs) on ONE GPU: the same launch path at 8x the headline grid.
    Determinism, permutation equivariance, finiteness, the 13-point budget - and one-step oracle parity
    on 64 of those states (sampled across the contact-count range)."""
    n = 32768
    v = make_vec(n)
    lo = torch.tensor(model["q_lower"][model["obs_order"]], dtype=torch.float32, device=DEV)
    hi = torch.tensor(model["q_upper"][model["obs_order"]], dtype=torch.float32, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(11)
    v.reset_tensor()
    for t in range(60):      # through touchdown: contact counts from 0 up to the budget
        v.step_tensor((lo + (hi - lo) * torch.rand(n, 25, device=DEV, generator=g)).contiguous())
    st = v.get_state().clone()
    a = (lo + (hi - lo) * torch.rand(n, 25, device=DEV, generator=g)).contiguous()
    o1, r1, _ = v.step_tensor(a)
    o1, r1, s1 = o1.clone(), r1.clone(), v.get_state().clone()
    cnt = torch.zeros(n, dtype=torch.int32, device=DEV)
    v.batch.contact_stats(cnt, None)
    assert torch.isfinite(o1).all() and torch.isfinite(r1).all() and torch.isfinite(s1).all()
    assert 0 <= int(cnt.min()) and int(cnt.max()) <= 13 and int(cnt.max()) >= 6
    v.set_state(st)                                   # determinism
    o2, r2, _ = v.step_tensor(a)
    assert (o2 == o1).all() and (r2 == r1).all() and (v.get_state() == s1).all()
    perm = torch.randperm(n, device=DEV, generator=g)   # permutation equivariance
    v.set_state(st[perm].contiguous())
    o3, r3, _ = v.step_tensor(a[perm].contiguous())
    assert (o3 == o1[perm]).all() and (r3 == r1[perm]).all() and (v.get_state() == s1[perm]).all()
    # oracle parity on a 64-env sample: the 32 envs with the most contacts + 32 spread over the batch
    idx = torch.cat([torch.argsort(cnt, descending=True)[:32], torch.arange(0, n, n // 32, device=DEV)[:32]]).cpu().numpy()
    st_h, a_h, o_h, r_h, c_h = st.cpu().numpy(), a.cpu().numpy(), o1.cpu().numpy(), r1.cpu().numpy(), cnt.cpu().numpy()
    fallback, worst_ratio = [], 0.0
    for e in idx:
        s = oracle64.new_state()
        oracle64.set_state(s, st_h[e].astype(np.float64))
        o, r, _ = oracle64.step(s, a_h[e].astype(np.float64))
        # These are the wildest states of the suite (60 random-action steps, up to 13 contact points, joint
        # rates of tens of rad/s). The angle error is the time integral of the rate error over the env-step
        # (5 x 0.002 s), so its bound is derived from the stated rate tolerance, not fitted:
        #   |dq| <= 1e-4 + 0.01 s * 5e-3 * max(1, |qd|_inf)
        qtol = 1e-4 + 0.01 * 5e-3 * max(1.0, np.abs(o[25:50]).max())
        try:
            assert_step_close(o_h[e], o, r_h[e], r, "config-4 env %d (%d contacts)" % (e, c_h[e]), q_atol=qtol)
        except AssertionError:
            # An ill-conditioned state (a 5 kg toe under kN contact forces, 60 unconverged sweeps): f32 itself is
            # the limit there. The yardstick is the oracle's OWN f32 build against its f64 build on this state -
            # the kernel (another f32 evaluation, different operation order) must stay within 3x that spread.
            # The stated tolerance stays mandatory for every state whose f32 spread is small, and the number of
            # states that may take this way out is bounded below.
            # (The scale of "what f32 does to this state" is estimated from SEVEN evaluations of the f32 build - the
            # state as it is and six copies with every entry moved by about one f32 ulp - not from one: a single
            # f32 evaluation can be lucky in the component that matters, and the sampled states are the kernel's own
            # trajectory, so they change with every kernel build.)
            prng = np.random.default_rng(int(e))
            spread = np.zeros(50)
            for k in range(7):
                s32 = oracle32.new_state()
                pert = st_h[e].astype(np.float64) * (1.0 + (6e-8 * prng.standard_normal(st_h[e].shape) if k else 0.0))
                oracle32.set_state(s32, pert)
                o32, r32, _ = oracle32.step(s32, a_h[e].astype(np.float64))
                spread = np.maximum(spread, np.abs(o32[:50] - o[:50]))
            assert spread[25:].max() > 1e-3 * max(1.0, np.abs(o[25:50]).max()), "well-conditioned state out of tolerance"
            err = np.abs(o_h[e][:50] - o[:50])
            assert (err <= 3 * spread + 1e-6).all(), "config-4 env %d beyond 3x the f32 spread" % e
            fallback.append(int(e))
            worst_ratio = max(worst_ratio, float((err / (spread + 1e-6)).max()))
        assert len(oracle64.contacts(s)[0]) == c_h[e]
    print("config-4 sample: %d of %d states judged by the f32-spread yardstick %s, worst error / spread %.2f"
          % (len(fallback), len(idx), fallback, worst_ratio))
    assert len(fallback) <= 3, "too many states outside the stated tolerance: %s" % fallback


@pytest.mark.parametrize("n", [1, 3, 63, 4095])
def test_ragged_batch_sizes(n):
    v = make_vec(n)
    obs = v.reset()
    ref = make_vec(2).reset()
    assert obs.shape == (n, 75) and np.array_equal(obs, np.tile(ref[0], (n, 1)))
    o, r, d, i = v.step(np.zeros((n, 25), np.float32))
    assert o.shape == (n, 75) and r.shape == (n,) and d.shape == (n,) and len(i) == n
    assert np.array_equal(o, np.tile(o[0], (n, 1)))


def test_time_limit_auto_reset():
    v = make_vec(4, max_episode_steps=3)
    first = v.reset().copy()
    for t in range(3):
        obs, rew, done, _ = v.step(np.zeros((4, 25), np.float32))
    assert done.all() and np.array_equal(obs, first)      # VecEnv semantics: obs of the new episode
    obs, rew, done, _ = v.step(np.zeros((4, 25), np.float32))
    assert not done.any()


def test_episode_limit_inside_the_step_launch():
    """trex_batch_set_episode_limit: the env whose count reaches the limit is reset BY THE STEP LAUNCH - reward of the
    finished step, done = 1, observation of the new episode (baselines' VecEnv semantics) - and equals an explicit
    reset bitwise; staggered counts end the episodes in different steps; the other envs are untouched."""
    a = torch.zeros(4, 25, device=DEV)
    v = make_vec(4, max_episode_steps=5)
    first = v.reset_tensor().clone()
    v.set_episode_steps(torch.tensor([4, 3, 0, 0], dtype=torch.int32))
    ref = make_vec(4)                      # no limit: the same physics without resets
    ref.reset_tensor()
    o, r, d = v.step_tensor(a)
    o0, r0, _ = ref.step_tensor(a)
    assert d.tolist() == [True, False, False, False]
    assert (o[0] == first[0]).all()                                   # env 0: first observation of its new episode
    assert (r == r0).all() and (o[1:] == o0[1:]).all()                # its reward is that of the finished step
    assert v.episode_steps.tolist() == [0, 4, 1, 1]
    o, r, d = v.step_tensor(a)
    o0, r0, _ = ref.step_tensor(a)
    assert d.tolist() == [False, True, False, False]
    assert (o[1] == first[1]).all() and (r[1:] == r0[1:]).all() and (o[2:] == o0[2:]).all()
    fresh = make_vec(4)                    # env 0 is now one step into a fresh episode
    fresh.reset_tensor()
    of, rf, _ = fresh.step_tensor(a)
    assert (o[0] == of[0]).all() and (r[0] == rf[0]).all()
    assert (v.get_state()[0] == fresh.get_state()[0]).all()


def test_non_finite_env_is_contained():
    """A NaN / Inf state resets that env (done = 1, reward 0) and leaves every other env bitwise untouched."""
    v = make_vec(6)
    v.reset()
    a = torch.zeros(6, 25, device=DEV)
    for _ in range(3):
        v.step_tensor(a)
    st = v.get_state().clone()
    ref = make_vec(6)
    ref.reset()
    ref.set_state(st)
    bad = st.clone()
    bad[2, 40] = float("nan")
    bad[5, 2] = float("inf")
    v.set_state(bad)
    o1, r1, d1 = v.step_tensor(a)
    o0, r0, d0 = ref.step_tensor(a)
    assert d1.tolist() == [False, False, True, False, False, True]
    assert torch.isfinite(o1).all() and torch.isfinite(r1).all() and torch.isfinite(v.get_state()).all()
    keep = [0, 1, 3, 4]
    assert (o1[keep] == o0[keep]).all() and (r1[keep] == r0[keep]).all()
    assert r1[2].item() == 0.0 and (v.get_state()[2, :3].cpu() == torch.tensor([0.0, 0.0, 3.0])).all()
    o2, r2, d2 = v.step_tensor(a)
    assert not d2.any()


def test_error_paths(capi):
    v = make_vec(2)
    with pytest.raises(ValueError):
        v.step_tensor(torch.zeros(2, 24, device=DEV))
    with pytest.raises(capi.TrexError):
        make_vec(2, starting_configuration={"no_such_joint": 0.1})
    with pytest.raises(capi.TrexError):
        capi.Batch(v.model, 0)


def test_single_env_facade_keeps_the_reference_surface(oracle64):
    from trex_gym import trex_env
    env = trex_env.TrexBulletEnv(ASSET_URDF)
    assert env.action_space.shape == (25,) and env.observation_space.shape == (75,)
    assert env.action_space.dtype == np.float32
    assert len(env.model._revolute_joint_indices) == 25
    assert abs(env.model._total_mass - 4834.87) < 0.01        # trex_train.py:123 logs this
    lo, hi = env.model.get_action_limits()
    assert np.allclose(env.action_space.low, lo) and np.allclose(env.observation_space.high[25:], 1e12)
    obs = env.reset()
    assert isinstance(obs, list) and len(obs) == 75
    s = oracle64.new_state()
    np.testing.assert_allclose(obs, oracle64.reset(s), atol=2e-6)
    o, r, d, info = env.step(np.zeros(50))       # the docstring's 2J-long action: first J entries used
    assert isinstance(o, list) and isinstance(r, float) and d is False and info == {}
    oo, ro, _ = oracle64.step(s, np.zeros(25))
    assert_step_close(np.array(o), oo, r, ro, "facade")
    assert env.seed(7) == [7]
    with pytest.raises(ValueError):
        env.step(np.zeros(3))


def test_episode_statistics_in_the_infos_of_the_numpy_api():
    """What bench.Monitor adds in the reference (trex_train.py:41-42): info['episode'] = {'r', 'l', 't'} on the step that
    ends an env's episode - here the harness's time limit inside the step launch. r is the sum of the rewards the caller
    was handed since the env's last reset (f64, like Monitor's Python floats), l the number of steps."""
    n, limit = 5, 4
    v = make_vec(n, max_episode_steps=limit)
    v.reset()
    v.set_episode_steps(torch.tensor([0, 1, 2, 3, 0], dtype=torch.int32))     # staggered ages: ends at steps 4, 3, 2, 1, 4
    rng = np.random.default_rng(3)
    lo, hi = v.action_space.low, v.action_space.high
    ret, length = np.zeros(n), np.zeros(n, int)
    seen = []
    for t in range(9):
        obs, rew, done, infos = v.step(rng.uniform(lo, hi, size=(n, 25)).astype(np.float32))
        ret += rew; length += 1
        for i in range(n):
            if done[i]:
                ep = infos[i]["episode"]
                assert ep["l"] == length[i] and abs(ep["r"] - ret[i]) < 1e-5 * max(1.0, abs(ret[i])) and ep["t"] >= 0.0
                seen.append((t, i, ep["l"]))
                ret[i], length[i] = 0.0, 0
            else:
                assert infos[i] == {}
    # first episodes are cut short by the staggered ages, the following ones last `limit` steps
    assert [(t, i) for t, i, _ in seen if t < 4] == [(0, 3), (1, 2), (2, 1), (3, 0), (3, 4)]
    assert all(l == limit for t, i, l in seen if t >= 4) and len(seen) == 5 + 6       # (env 3 ends a third time at t = 8)


@pytest.mark.parametrize("K", [5, 10])
def test_k_steps_through_contact_separate_no_faster_than_the_f32_oracle(K, oracle64, oracle32, model):
    """VERDICT r3 item 6a. One-step parity says nothing about how fast the kernel and the f64 oracle drift apart over a few
    steps of CONTACT (trajectories are chaotic there: any two f32 evaluations separate). The yardstick is the oracle's own
    f32 build: from the 50 landing states (free fall, impact, standing under noise), K steps of one action sequence on the GPU,
    on the f64 oracle and on the f32 oracle. Per state the kernel's separation from f64 is at most 3 x the f32 oracle's (+ a
    floor where both are at rounding level: 2e-5 rad, 1e-4 of the rate scale), and over the 50 states the kernel is the CLOSER
    of the two (median ratio < 1; measured 0.4 - 0.5, scripts/kstep_separation.py)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "scripts"))
    from kstep_separation import landing_states, separations
    states = landing_states(oracle64, model)
    assert len(states) == 50
    s = separations(K, states, model, oracle64, oracle32, DEV)
    dq_g, dq_o, dqd_g, dqd_o = s[:, 0], s[:, 1], s[:, 2], s[:, 3]
    assert np.all(dq_g <= 3.0 * dq_o + 2e-5), (dq_g - 3.0 * dq_o).max()
    assert np.all(dqd_g <= 3.0 * dqd_o + 1e-4), (dqd_g - 3.0 * dqd_o).max()
    assert np.median(dq_g / np.maximum(dq_o, 1e-12)) < 1.0 and np.median(dqd_g / np.maximum(dqd_o, 1e-12)) < 1.0
    assert dq_g.max() <= 3.0 * dq_o.max() and dqd_g.max() <= 3.0 * dqd_o.max()
    assert dqd_o.max() > 1e-4       # the states do amplify rounding: the comparison is not vacuous


def test_config5_at_4096_envs_is_deterministic_and_permutation_equivariant():
    """BASELINE config 5 at its full size: per-env mass scales and friction, 4096 envs, 40 steps from a
    staggered landing. Two runs give the same bits; a batch with its envs PERMUTED - state, actions, mass scales and friction
    permuted along - gives the permuted rows, bitwise (one env per wave: no env sees another, whatever slot or SIMD it runs on
    and whatever the rank lists hand it)."""
    n, steps = 4096, 40
    g = torch.Generator(device=DEV).manual_seed(1)
    ms = 0.8 + 0.4 * torch.rand(n, 26, device=DEV, generator=g)
    mu = 0.5 + 0.75 * torch.rand(n, device=DEV, generator=g)
    perm = torch.randperm(n, device=DEV, generator=g)
    lo = torch.as_tensor(make_vec(1).action_space.low, device=DEV)
    hi = torch.as_tensor(make_vec(1).action_space.high, device=DEV)
    acts = lo + (hi - lo) * torch.rand(steps, n, 25, device=DEV, generator=g)

    def run(order):
        v = make_vec(n, max_episode_steps=30)
        v.set_domain(ms[order], mu[order])
        v.reset_tensor()
        v.set_episode_steps((order % 30).to(torch.int32))       # episodes end inside the launches, at env-specific steps
        out = []
        for t in range(steps):
            v.step_tensor(acts[t][order])
            out.append(v.rows.clone())
        return torch.stack(out), v.get_state()

    ident = torch.arange(n, device=DEV)
    rows_a, st_a = run(ident)
    rows_b, st_b = run(ident)
    assert torch.equal(rows_a, rows_b) and torch.equal(st_a, st_b)                      # determinism
    rows_p, st_p = run(perm)
    assert torch.equal(rows_p, rows_a[:, perm]) and torch.equal(st_p, st_a[perm])       # equivariance
    assert torch.isfinite(rows_a).all()
    cnt = (rows_a[-1][:, 50:75].abs() > 0).any(1).sum().item()
    assert cnt > n // 2          # motors are on and most envs have landed: the run exercises contact rows
    assert rows_a[:, :, 76].sum().item() >= n                                             # every env's episode ended at least once


def test_config5_randomised_states_against_the_oracle(oracle64, oracle32, model):
    """32 states spread over free fall, landing and rest, each with its own body-mass scales U(0.8, 1.2) and friction
    U(0.5, 1.25) (the ranges of BASELINE config 5): one env-step on the GPU against the f64 oracle with the same domain."""
    rng = np.random.default_rng(7)
    q0 = model["q_start"][model["obs_order"]]
    lo, hi = model["q_lower"][model["obs_order"]], model["q_upper"][model["obs_order"]]
    n = 32
    ms = rng.uniform(0.8, 1.2, (n, 26)).astype(np.float32)
    mu = rng.uniform(0.5, 1.25, n).astype(np.float32)
    states, acts, kinds = [], [], []
    for e in range(n):
        # the env's OWN trajectory under its own domain, stopped in free fall (steps 3..20), around touchdown (26..60) or at rest (150..300)
        kind = ("airborne", "landing", "rest")[e % 3]
        stop = {"airborne": rng.integers(3, 21), "landing": rng.integers(26, 61), "rest": rng.integers(150, 301)}[kind]
        s = oracle64.new_state()
        oracle64.set_domain(s, ms[e].astype(np.float64), float(mu[e]))
        oracle64.reset(s)
        for t in range(int(stop)):
            oracle64.step(s, np.clip(q0 + 0.2 * rng.normal(size=25), lo, hi))
        states.append(oracle64.get_state(s).astype(np.float32))
        acts.append(rng.uniform(lo, hi).astype(np.float32))
        kinds.append(kind)
    states, acts = np.array(states), np.array(acts)
    v = make_vec(n)
    v.reset()
    v.set_domain(torch.tensor(ms), torch.tensor(mu))
    v.set_state(torch.tensor(states))
    obs, rew, _, _ = v.step(acts)
    cnt = torch.zeros(n, dtype=torch.int32, device=DEV)
    v.batch.contact_stats(cnt, None)
    in_contact, fallback = 0, []
    for e in range(n):
        s = oracle64.new_state()
        oracle64.set_domain(s, ms[e].astype(np.float64), float(mu[e]))
        oracle64.set_state(s, states[e].astype(np.float64))
        o, r, _ = oracle64.step(s, acts[e].astype(np.float64))
        qtol = 1e-4 + 0.01 * 3e-3 * max(1.0, np.abs(o[25:50]).max())      # (the angle error integrates the rate error over the env-step)
        try:
            assert_step_close(obs[e], o, rew[e], r, "config-5 %s state %d" % (kinds[e], e), q_atol=qtol)
        except AssertionError:
            # the way out of test_config4_size_on_one_gpu, bounded the same way: a state on which f32 ITSELF is the limit (a
            # rex at rest hit by a full-range target jump: saturated motors on 2 kg toes under kN contact forces) is judged
            # by 3 x the spread of the oracle's own f32 build (seven evaluations, the state moved by an f32 ulp) against f64
            prng = np.random.default_rng(e)
            spread = np.zeros(50)
            for k in range(7):
                s32 = oracle32.new_state()
                oracle32.set_domain(s32, ms[e].astype(np.float64), float(mu[e]))
                oracle32.set_state(s32, states[e].astype(np.float64) * (1.0 + (6e-8 * prng.standard_normal(states[e].shape) if k else 0.0)))
                o32, _, _ = oracle32.step(s32, acts[e].astype(np.float64))
                spread = np.maximum(spread, np.abs(o32[:50] - o[:50]))
            assert spread[25:].max() > 1e-3 * max(1.0, np.abs(o[25:50]).max()), "well-conditioned state out of tolerance: %s %d" % (kinds[e], e)
            assert (np.abs(obs[e][:50] - o[:50]) <= 3 * spread + 1e-6).all(), "config-5 state %d beyond 3x the f32 spread" % e
            fallback.append(e)
        nco = len(oracle64.contacts(s)[0])
        assert cnt[e].item() == nco, (e, kinds[e])
        in_contact += nco > 0
    print("config-5 sample: %d of %d states judged by the f32-spread yardstick %s" % (len(fallback), n, fallback))
    assert len(fallback) <= 3
    assert 12 <= in_contact <= 28           # all three regimes are present
