"""Physics invariants that pin the CPU oracle's dynamics (oracle/trex_oracle.c) independently of
pybullet (absent; parity unpinned - SURVEY 8c):
  * M^-1 from the ABA delta sweeps == inverse of a mass matrix assembled from geometric
    Jacobians about each body's COM (an independent formulation, numpy only);
  * momentum and energy behaviour in free flight;
  * complementarity / bounds of the PGS solution; weight carried at rest; joint limits.
"""
import numpy as np
import pytest

from conftest import ASSET_URDF
from oracle import oracle as O
from oracle import trex_model as tm


def random_state(model, rng, vel=1.0, height=5.0):
    nj = model["nb"] - 1
    st = np.zeros(13 + 2 * nj)
    st[0:3] = [rng.normal(), rng.normal(), height]
    q = rng.normal(size=4)
    st[3:7] = q / np.linalg.norm(q)
    st[7:13] = vel * rng.normal(size=6)
    lo, hi = model["q_lower"][model["obs_order"]], model["q_upper"][model["obs_order"]]
    st[13:13 + nj] = rng.uniform(0.8 * lo, 0.8 * hi)
    st[13 + nj:] = vel * rng.normal(size=nj)
    return st


def mass_matrix_from_jacobians(model, orc, s):
    """M = sum_b Jv^T m Jv + Jw^T Ic Jw with per-body COM Jacobians, generalised velocity
    [w_base, v_base_origin, qd (body order)]."""
    pos, rot = orc.body_poses(s)
    nb = model["nb"]
    nd = 6 + nb - 1
    M = np.zeros((nd, nd))
    axis_w = [rot[i] @ model["joint_axis"][i] for i in range(nb)]
    for b in range(nb):
        c = pos[b] + rot[b] @ model["com"][b]
        a = model["inertia"][b]
        Ib = np.array([[a[0], a[1], a[2]], [a[1], a[3], a[4]], [a[2], a[4], a[5]]])
        Ic = rot[b] @ Ib @ rot[b].T
        Jw, Jv = np.zeros((3, nd)), np.zeros((3, nd))
        Jw[:, 0:3] = np.eye(3)
        Jv[:, 3:6] = np.eye(3)
        d = c - pos[0]
        Jv[:, 0:3] = -np.array([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]])  # w x d
        i = b
        while i >= 1:
            Jw[:, 6 + i - 1] = axis_w[i]
            Jv[:, 6 + i - 1] = np.cross(axis_w[i], c - pos[i])
            i = model["parent"][i]
        M += model["mass"][b] * Jv.T @ Jv + Jw.T @ Ic @ Jw
    return M


def test_minv_against_independent_mass_matrix(model, oracle64):
    rng = np.random.default_rng(0)
    for _ in range(5):
        s = oracle64.new_state()
        oracle64.set_state(s, random_state(model, rng))
        M = mass_matrix_from_jacobians(model, oracle64, s)
        Minv = oracle64.minv(s)
        np.testing.assert_allclose(Minv, Minv.T, rtol=0, atol=1e-12)
        np.testing.assert_allclose(Minv @ M, np.eye(M.shape[0]), atol=2e-9)


def _free_oracle(model, **over):
    m = dict(model)
    m["joint_damping"] = np.zeros_like(model["joint_damping"])
    p = dict(link_damping=0.0, max_coordinate_velocity=1e9)
    p.update(over)
    return O.Oracle(m, params=p)


def test_free_flight_momentum(model):
    """No contact, motors off: d(linear momentum)/dt = -M g and angular momentum about the COM is
    conserved; the residual is the first-order integrator error, so it must shrink ~linearly with dt
    (measured: exactly 4x per 4x dt)."""
    mtot = model["mass"].sum()
    errs = []
    for dt in (5e-4, 1.25e-4):
        orc = _free_oracle(model, dt=dt)
        rng = np.random.default_rng(1)
        s = orc.new_state()
        orc.set_state(s, random_state(model, rng, vel=0.5, height=50.0))

        def com_momenta(s):
            e = orc.energy(s)
            pos, rot = orc.body_poses(s)
            com = sum(model["mass"][b] * (pos[b] + rot[b] @ model["com"][b]) for b in range(model["nb"])) / mtot
            h = e["momentum"]
            lin = h[3:6]
            return lin, h[0:3] - np.cross(com - pos[0], lin)
        l0, a0 = com_momenta(s)
        n = int(round(0.1 / dt))
        for _ in range(n):
            orc.substep(s)
        l1, a1 = com_momenta(s)
        errs.append((np.abs(l1 - l0 - [0, 0, -mtot * 9.81 * n * dt]).max() / (mtot * 0.5),
                     np.abs(a1 - a0).max() / max(1.0, np.abs(a0).max())))
    assert errs[1][0] < 1e-5 and errs[1][1] < 1e-4
    assert errs[1][0] < 0.3 * errs[0][0] and errs[1][1] < 0.3 * errs[0][1]


def test_energy_conservation_scales_with_dt(model):
    """Zero gravity, no damping: kinetic energy drift of the semi-implicit scheme is O(dt)."""
    drift = []
    for dt in (4e-4, 1e-4):
        orc = _free_oracle(model, gravity=0.0, dt=dt)
        rng = np.random.default_rng(2)
        s = orc.new_state()
        st = random_state(model, rng, vel=0.3, height=50.0)
        orc.set_state(s, st)
        k0 = orc.energy(s)["ke"]
        for _ in range(int(round(0.05 / dt))):
            orc.substep(s)
            assert orc.limit_rows(s) == 0
        drift.append(abs(orc.energy(s)["ke"] - k0) / k0)
    assert drift[1] < 5e-3
    assert drift[1] < 0.5 * drift[0]


def test_forward_dynamics_consistent_with_minv(model, oracle64):
    """qdd(tau1) - qdd(tau0) = Minv[joint rows] @ [0; tau1 - tau0] (linearity in tau)."""
    rng = np.random.default_rng(3)
    s = oracle64.new_state()
    oracle64.set_state(s, random_state(model, rng))
    t0, t1 = rng.normal(size=25) * 100, rng.normal(size=25) * 100
    q0, b0 = oracle64.forward_dynamics(s, t0)
    q1, b1 = oracle64.forward_dynamics(s, t1)
    Minv = oracle64.minv(s)
    f = np.zeros(31)
    order = model["obs_order"]
    f[6 + order - 1] = t1 - t0
    dv = Minv @ f
    np.testing.assert_allclose(q1 - q0, dv[6 + order - 1], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(b1 - b0, dv[:6], rtol=1e-9, atol=1e-9)


def test_reset_semantics(model, oracle64):
    """trex_env.py:98-122: start pose, one un-actuated settle substep, obs = q, qd, tau=0."""
    s = oracle64.new_state()
    obs = oracle64.reset(s)
    q0 = model["q_start"][model["obs_order"]]
    np.testing.assert_allclose(obs[50:], 0)
    assert np.abs(obs[:25] - q0).max() < 1e-3     # 2 ms of free fall moves joints very little
    st = oracle64.get_state(s)
    # free fall for one dt with link damping: v_z = -g*dt (to first order)
    assert abs(st[9] + 9.81 * 0.002) < 1e-3
    assert abs(st[2] - (3.0 - 9.81 * 0.002 ** 2)) < 1e-4


def test_step_bounds_and_complementarity(model, oracle64):
    orc = oracle64
    rng = np.random.default_rng(4)
    s = orc.new_state()
    orc.reset(s)
    lo, hi = model["q_lower"][model["obs_order"]], model["q_upper"][model["obs_order"]]
    q0 = model["q_start"][model["obs_order"]]
    seen_contact = False
    for i in range(150):
        a = np.clip(q0 + 0.2 * rng.normal(size=25), lo, hi)
        obs, r, pen = orc.step(s, a * 3.0)   # out-of-range actions are clipped (trex_env.py:147)
        assert np.all(np.abs(obs[50:]) <= 3e5 * (1 + 1e-9))
        b, lam, pos, dist = orc.contacts(s)
        if len(b):
            seen_contact = True
            assert np.all(lam[:, 0] >= 0)
            assert np.all(np.abs(lam[:, 1]) <= 0.25 * lam[:, 0] + 1e-9)
            assert np.all(np.abs(lam[:, 2]) <= 0.25 * lam[:, 0] + 1e-9)
        assert np.isfinite(obs).all() and np.isfinite(r)
        np.testing.assert_allclose(r, -pen.sum(), rtol=1e-12)
    assert seen_contact


def test_crouch_comes_to_rest_and_ground_carries_weight(model, oracle64):
    """Holding the start pose, the animal lands and the summed normal force settles at m*g."""
    orc = oracle64
    s = orc.new_state()
    orc.reset(s)
    q0 = model["q_start"][model["obs_order"]]
    fz = []
    for i in range(400):
        orc.step(s, q0)
        b, lam, pos, dist = orc.contacts(s)
        fz.append(lam[:, 0].sum() / 0.002 if len(b) else 0.0)
    st = orc.get_state(s)
    w = model["mass"].sum() * 9.81
    assert abs(np.mean(fz[-50:]) - w) < 0.05 * w
    assert np.abs(st[7:13]).max() < 0.2
    assert 0.5 < st[2] < 3.0
    # it comes to rest as a tripod: toes (+ metatarsus) and the tail tip
    assert all(any(k in model["body_names"][i] for k in ("toe", "tarsometatarsus", "caudal")) for i in b)


def test_joint_limit_rows(model, oracle64):
    orc = oracle64
    s = orc.new_state()
    st = np.zeros(63)
    st[0:3] = [0, 0, 50]
    st[6] = 1
    lo = model["q_lower"][model["obs_order"]]
    st[13:38] = lo - 0.05        # every joint 0.05 rad past its lower stop
    orc.set_state(s, st)
    orc.set_motors_on(s, 0)
    orc.substep(s)
    assert orc.limit_rows(s) == 25
    qd = orc.get_state(s)[38:63]
    # ERP 0.2 asks each joint for >= 0.2*0.05/dt = 5 rad/s. 60 PGS sweeps are not converged on the
    # ill-conditioned neck chain (947 kg cranium behind 12 kg atlas), so most - not all - are there;
    assert np.sum(np.abs(qd - 5.0) < 0.1) >= 20
    # the converged solution satisfies every unilateral row.
    orc.set_param("iterations", 6000)
    orc.set_state(s, st)
    orc.substep(s)
    orc.set_param("iterations", 60)
    assert np.all(orc.get_state(s)[38:63] > 5.0 - 1e-2)


def test_f32_oracle_tracks_f64(model, oracle64, oracle32):
    s64, s32 = oracle64.new_state(), oracle32.new_state()
    oracle64.reset(s64); oracle32.reset(s32)
    a = model["q_start"][model["obs_order"]]
    for i in range(10):   # contact-free window
        o64, r64, _ = oracle64.step(s64, a)
        o32, r32, _ = oracle32.step(s32, a)
    np.testing.assert_allclose(o32[:50], o64[:50], atol=2e-4)
    np.testing.assert_allclose(o32[50:], o64[50:], atol=2e-2 * max(1.0, np.abs(o64[50:]).max()))
    assert abs(r32 - r64) < 1e-3 * abs(r64)


def _lowest_hull_vertex(model, orc, s):
    pos, rot = orc.body_poses(s)
    hs, hv = model["hull_start"], model["hull_xyz"]
    z = [(pos[b][2] + (rot[b].reshape(3, 3) @ hv[hs[b]:hs[b + 1]].T)[2]).min() for b in range(model["nb"]) if hs[b + 1] > hs[b]]
    return min(z)


def _settle(model, steps=400, **params):
    orc = O.Oracle(model, params=params)
    s = orc.new_state()
    orc.reset(s)
    q0 = model["q_start"][model["obs_order"]]
    bodies = 0
    for _ in range(steps):
        orc.step(s, q0)
        bodies = max(bodies, len(set(orc.contacts(s)[0].astype(int))))
    lam = orc.contacts(s)[1]
    weight = lam[:, 0].sum() / 0.002 / (model["mass"].sum() * 9.81)
    return orc, s, orc.get_state(s), bodies, weight


def test_contact_budget_degrades_gracefully(model):
    """More touching bodies than contact rows: the budget goes to the DEEPEST bodies (oracle and kernel alike), so
    no body sinks for lack of a row. Quantified against an effectively unbudgeted run (64 points):
      * the headline budget (13) at the standard margin: rest height within 0.5 mm, ground carries the weight to 1 %
      * 17 bodies inside an inflated 0.5 m margin, budget 13: the same rest pose to 0.5 mm, nothing below the floor
      * budget 4 with 7 bodies touching: the supporting set rotates (jitter: weight within 35 %), yet nothing sinks
        (lowest vertex within 5 mm of the floor) and the rest height stays within 5 mm."""
    floor = 0.0005
    _, _, ref, bodies64, w64 = _settle(model, max_contacts=64)
    assert bodies64 >= 7 and abs(w64 - 1) < 0.01
    orc, s, st, _, w = _settle(model, max_contacts=13)
    assert abs(st[2] - ref[2]) < 5e-4 and abs(w - 1) < 0.01
    orc, s, st, bodies, w = _settle(model, max_contacts=13, contact_margin=0.5)
    assert len(orc.contacts(s)[0]) == 13                       # budget exhausted, one point per kept body
    orc64, s64, st64, bodies_all, _ = _settle(model, max_contacts=64, contact_margin=0.5)
    assert bodies_all > 13                                       # more bodies inside the margin than rows
    assert abs(st[2] - st64[2]) < 5e-4 and abs(w - 1) < 0.01
    assert _lowest_hull_vertex(model, orc, s) > floor - 1e-3
    orc, s, st, bodies, w = _settle(model, max_contacts=4)
    assert len(orc.contacts(s)[0]) == 4
    assert _lowest_hull_vertex(model, orc, s) > floor - 5e-3
    assert abs(st[2] - ref[2]) < 5e-3 and abs(w - 1) < 0.35
