"""The trainer-side oracle (oracle/ppo_oracle.py: f64 numpy restatement of GAE, the PPO2 losses, Adam and
VecNormalize's running statistics - parity unpinned: baselines / TensorFlow are absent, see its header) checked
against itself: analytic gradients vs central differences, closed forms, and plain numpy on the pooled data."""
import os

import numpy as np
import pytest

from oracle import ppo_oracle as P


def _params(rng, obs_dim=7, act_dim=3, hidden=5):
    p = {}
    for pre, out in (("pi", act_dim), ("vf", 1)):
        for k, (o, i) in zip((0, 2, 4), ((hidden, obs_dim), (hidden, hidden), (out, hidden))):
            p["%s.%d.weight" % (pre, k)] = rng.normal(size=(o, i)) * 0.5
            p["%s.%d.bias" % (pre, k)] = rng.normal(size=o) * 0.1
    p["logstd"] = rng.normal(size=act_dim) * 0.3
    return p


def test_running_mean_std_equals_pooled_statistics():
    rng = np.random.default_rng(0)
    batches = [rng.normal(2.0, 3.0, size=(n, 4)) for n in (5, 1, 64, 17)]
    r = P.RunningMeanStd((4,))
    for b in batches:
        r.update(b)
    pooled = np.concatenate(batches)
    # the 1e-4 pseudo-count of the initial (mean 0, var 1) state is all that separates the two
    np.testing.assert_allclose(r.mean, pooled.mean(0), rtol=1e-5)
    np.testing.assert_allclose(r.var, pooled.var(0), rtol=1e-4)
    assert abs(r.count - (len(pooled) + 1e-4)) < 1e-12


def test_vecnormalize_clips_and_discounts():
    vn = P.VecNormalize(3, 2, gamma=0.5)
    o = vn.obs(np.array([[1.0, 100.0], [2.0, -100.0], [3.0, 0.0]]))
    assert np.abs(o).max() <= 10.0 and abs(o[:, 0].mean()) < 1e-3
    r1 = vn.reward(np.array([1.0, 1.0, 1.0]), np.array([False, True, False]))
    assert vn.ret.tolist() == [1.0, 0.0, 1.0]                 # the finished env's return restarts
    vn.reward(np.array([1.0, 1.0, 1.0]), np.array([False, False, False]))
    assert vn.ret.tolist() == [1.5, 1.0, 1.5]
    assert (np.abs(r1) <= 10.0).all()


def test_gae_closed_forms():
    T, n = 6, 2
    rew = np.ones((T, n)); val = np.zeros((T, n)); last = np.zeros(n); dones = np.zeros((T, n))
    adv, ret = P.gae(rew, val, last, dones, gamma=0.9, lam=1.0)
    # lam = 1, V = 0: the advantage is the discounted reward-to-go
    want = np.array([sum(0.9 ** k for k in range(T - t)) for t in range(T)])
    np.testing.assert_allclose(adv[:, 0], want, rtol=1e-14)
    # lam = 0: one-step TD error
    val = np.arange(T * n, dtype=float).reshape(T, n)
    adv0, _ = P.gae(rew, val, np.array([7.0, 8.0]), dones, gamma=0.9, lam=0.0)
    nextv = np.vstack([val[1:], [7.0, 8.0]])
    np.testing.assert_allclose(adv0, rew + 0.9 * nextv - val, rtol=1e-14)
    # an episode end cuts both the bootstrap and the recursion
    dones[2, 0] = 1.0
    adv1, ret1 = P.gae(rew, val, np.array([7.0, 8.0]), dones, gamma=0.9, lam=0.95)
    assert abs(adv1[2, 0] - (1.0 - val[2, 0])) < 1e-14
    np.testing.assert_allclose(ret1, adv1 + val)


@pytest.mark.parametrize("ent_coef", [0.0, 0.01])
def test_loss_gradients_match_central_differences(ent_coef):
    rng = np.random.default_rng(1)
    p = _params(rng)
    n = 40
    obs = rng.normal(size=(n, 7)); noise = rng.normal(size=(n, 3))
    mean, logstd, v0 = P.policy_forward(p, obs)
    act = P.sample_action(mean, logstd, noise)
    nlp0 = P.neglogp(mean, logstd, act) + 0.3 * rng.normal(size=n)     # old policy differs: ratios spread past the clip
    v_old = v0 + 0.3 * rng.normal(size=n)
    adv = rng.normal(size=n); ret = v_old + rng.normal(size=n)
    out, g = P.ppo_loss_and_grads(p, obs, act, nlp0, v_old, adv, ret, ent_coef=ent_coef)
    assert ((out["ratio"] > 1.2).any() or (out["ratio"] < 0.8).any()) and np.isfinite(out["loss"])
    for k in p:
        flat = p[k].reshape(-1)
        for i in rng.choice(flat.size, size=min(4, flat.size), replace=False):
            old = flat[i]
            h = 1e-6
            flat[i] = old + h; lp = P.ppo_loss_and_grads(p, obs, act, nlp0, v_old, adv, ret, ent_coef=ent_coef)[0]["loss"]
            flat[i] = old - h; lm = P.ppo_loss_and_grads(p, obs, act, nlp0, v_old, adv, ret, ent_coef=ent_coef)[0]["loss"]
            flat[i] = old
            np.testing.assert_allclose(g[k].reshape(-1)[i], (lp - lm) / (2 * h), rtol=2e-5, atol=1e-9, err_msg=k)


def test_adam_first_steps_and_global_norm_clip():
    p = {"w": np.array([1.0, -2.0])}
    g = {"w": np.array([0.3, -0.4])}                      # norm 0.5: not clipped
    gc, norm = P.clip_by_global_norm(g, 0.5)
    assert abs(norm - 0.5) < 1e-15 and np.allclose(gc["w"], g["w"])
    gc, _ = P.clip_by_global_norm({"w": np.array([3.0, -4.0])}, 0.5)
    assert abs(np.linalg.norm(gc["w"]) - 0.5) < 1e-15
    opt = P.Adam(p, lr=1e-2, epsilon=1e-5)
    p = opt.step(p, g)
    # t = 1: m = 0.1 g, v = 0.001 g^2, lr_t = lr sqrt(0.001) / 0.1 -> step = lr g / (|g| + eps / sqrt(0.001))
    want = np.array([1.0, -2.0]) - 1e-2 * g["w"] / (np.abs(g["w"]) + 1e-5 / np.sqrt(1e-3))
    np.testing.assert_allclose(p["w"], want, rtol=1e-12)


def test_running_moments_merge_across_ranks():
    """trex_gym.ppo.merge_running_moments (the data-parallel trainer's once-per-rollout exchange): every rank starts from
    the same statistics and folds in its own data; the merged statistics are those of the prefix data and every rank's new
    data ONCE - checked against numpy on the concatenation, for vectors (observations) and scalars (returns)."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "trex-gym_amd"))
    from trex_gym.ppo import merge_running_moments

    def chan(mean, var, count, x):      # RunningMeanStd.update (baselines): fold a batch in
        bm, bv, bn = x.mean(0), x.var(0), x.shape[0]
        delta, tot = bm - mean, count + bn
        return mean + delta * bn / tot, (var * count + bv * bn + delta ** 2 * count * bn / tot) / tot, tot

    rng = np.random.default_rng(0)
    D = 7
    base = rng.normal(2.0, 3.0, (500, D))
    prefix = chan(np.zeros(D), np.ones(D), 1e-4, base)
    parts, new = [], []
    for r, n in enumerate((4096, 4096, 37)):
        x = rng.normal(-1.0 + r, 0.5 + r, (n, D))
        new.append(x)
        parts.append(chan(*prefix, x))
    mean, var, cnt = merge_running_moments(prefix, parts)
    want = chan(*prefix, np.concatenate(new))
    np.testing.assert_allclose(mean, want[0], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(var, want[1], rtol=1e-10)
    assert abs(cnt - want[2]) < 1e-6
    # a rank that saw nothing new changes nothing; scalars work too
    m2, v2, c2 = merge_running_moments(prefix, [prefix, parts[0]])
    np.testing.assert_allclose(m2, parts[0][0], rtol=1e-12)
    sm, sv, sc = merge_running_moments((0.5, 2.0, 10.0), [chan(0.5, 2.0, 10.0, np.array([1.0, 3.0, 5.0])), chan(0.5, 2.0, 10.0, np.array([-2.0]))])
    w = chan(0.5, 2.0, 10.0, np.array([1.0, 3.0, 5.0, -2.0]))
    assert abs(sm - w[0]) < 1e-12 and abs(sv - w[1]) < 1e-12 and sc == 14.0
