"""The trainer-side HIP kernels (include/trex_policy.h, csrc/policy_step.hip) against the f64 oracle of the PPO2 /
VecNormalize arithmetic (oracle/ppo_oracle.py - parity unpinned, see its header) and against the same MlpPolicy
evaluated by plain f32 PyTorch. Tolerances: 1e-5 relative for everything computed in f32 on both sides of a short
chain (policy outputs, advantages, returns, losses), 1e-10 for the f64 running statistics, and for the parameter
update 1e-5 of the step size lr (Adam's first steps are +-lr per element whatever the gradient's size)."""
import math

import numpy as np
import pytest
import torch

from oracle import ppo_oracle as P

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _theta_to_params(kern, theta):
    """flat [in, out] vector -> the oracle's dict in nn.Sequential naming with [out, in] weights"""
    th = theta.detach().cpu().double().numpy()
    out = {}
    for name, (off, shape) in kern.layout.items():
        v = th[off:off + math.prod(shape)].reshape(shape)
        if name == "logstd":
            out["logstd"] = v
            continue
        net, what = name.split(".")
        k = {"1": 0, "2": 2, "3": 4}[what[1]]
        out["%s.%d.%s" % (net, k, "weight" if what[0] == "W" else "bias")] = v.T.copy() if what[0] == "W" else v
    return out


def _grads_to_flat(kern, grads):
    flat = np.zeros(kern.param_count)
    for name, (off, shape) in kern.layout.items():
        if name == "logstd":
            g = grads["logstd"]
        else:
            net, what = name.split(".")
            k = {"1": 0, "2": 2, "3": 4}[what[1]]
            g = grads["%s.%d.%s" % (net, k, "weight" if what[0] == "W" else "bias")]
            g = g.T if what[0] == "W" else g
        flat[off:off + math.prod(shape)] = np.asarray(g).reshape(-1)
    return flat


@pytest.mark.parametrize("n", [4096, 77])
def test_policy_kernel_matches_f32_torch_and_the_oracle(n):
    """trex_policy_act: normalise + clip, both MLPs on the matrix cores, sample, log-prob - against MlpPolicy in f32
    torch at 1e-5 and against the f64 oracle; ragged batch (n not a multiple of the 32-env tile) included."""
    from trex_gym import _capi
    from trex_gym.ppo import MlpPolicy
    torch.manual_seed(1)
    D, A = 75, 25
    kern = _capi.Policy(n, D, A, 64, 0)
    pol = MlpPolicy(kern.layout, kern.param_count, torch.device(DEV))
    with torch.no_grad():       # away from the special initial point (last layers at gain 0.01, biases 0, logstd 0)
        pol.theta.add_(0.05 * torch.randn_like(pol.theta))
    rows = torch.randn(n, 77, device=DEV) * torch.linspace(0.1, 30.0, 77, device=DEV)
    rows[:, 5] = 1e4            # a column that must clip
    st = dict(obs_mean=np.linspace(-1, 1, D), obs_var=np.linspace(0.5, 9.0, D), obs_count=100.0, ret_mean=0.0, ret_var=4.0,
              ret_count=100.0)
    kern.set_stats(st)
    noise = torch.randn(n, A, device=DEV)
    actions, obs_n, act_b = torch.empty(n, A, device=DEV), torch.empty(n, D, device=DEV), torch.empty(n, A, device=DEV)
    logp, val = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    kern.act(pol.theta, rows, noise, actions, obs_n, act_b, logp, val, clip_obs=10.0)
    # f32 torch reference of the same op
    mean32 = torch.tensor(st["obs_mean"], device=DEV, dtype=torch.float32)
    rstd32 = torch.tensor(1.0 / np.sqrt(st["obs_var"] + 1e-8), device=DEV, dtype=torch.float32)
    with torch.no_grad():
        o = ((rows[:, :D] - mean32) * rstd32).clamp(-10.0, 10.0)
        d = pol.dist(o)
        a = d.loc + d.scale * noise
        lp = d.log_prob(a).sum(-1)
        v = pol.value(o)
    assert float(o.abs().max()) == 10.0
    torch.testing.assert_close(obs_n, o, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(actions, a, rtol=1e-5, atol=1e-5)
    assert torch.equal(act_b, actions)
    torch.testing.assert_close(val, v, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(logp, lp, rtol=1e-5, atol=2e-4)      # sum of 25 squares of O(1): 1e-5 of its terms
    # f64 oracle
    prm = _theta_to_params(kern, pol.theta)
    o64 = np.clip((rows[:, :D].cpu().double().numpy() - st["obs_mean"]) / np.sqrt(st["obs_var"] + 1e-8), -10, 10)
    m64, ls64, v64 = P.policy_forward(prm, o64)
    a64 = P.sample_action(m64, ls64, noise.cpu().double().numpy())
    np.testing.assert_allclose(actions.cpu().numpy(), a64, rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(val.cpu().numpy(), v64, rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(-logp.cpu().numpy(), P.neglogp(m64, ls64, actions.cpu().double().numpy()), rtol=1e-5, atol=3e-4)
    # value_only writes nothing but the values
    val2, keep = torch.empty(n, device=DEV), actions.clone()
    kern.act(pol.theta, rows, None, None, value_out=val2, value_only=True)
    assert torch.equal(val2, val) and torch.equal(actions, keep)
    with pytest.raises(_capi.TrexError):
        kern.act(pol.theta, rows, noise.cpu(), actions)             # host tensor refused, no GPU fault


def test_observe_kernel_is_vecnormalize():
    """trex_policy_observe over several steps with episode ends == the oracle's VecNormalize (f64 state)."""
    from trex_gym import _capi
    n, D = 1000, 75          # ragged: 15.6 workgroups of 64 rows
    kern = _capi.Policy(n, D, 25, 64, 0)
    vn = P.VecNormalize(n, D, gamma=0.99)
    g = torch.Generator(device=DEV).manual_seed(2)
    rows = torch.empty(n, 77, device=DEV)
    rows[:, :D] = 5 + 3 * torch.randn(n, D, device=DEV, generator=g)
    kern.observe(rows, with_reward=False)
    vn.obs(rows[:, :D].cpu().double().numpy())
    raw, done, scale = torch.empty(n, device=DEV), torch.empty(n, device=DEV), torch.empty(1, device=DEV)
    for t in range(6):
        rows[:, :D] = (5 + t) + (3 + t) * torch.randn(n, D, device=DEV, generator=g)
        rows[:, D] = -100 * torch.rand(n, device=DEV, generator=g)
        rows[:, D + 1] = (torch.rand(n, device=DEV, generator=g) < 0.2).float()
        kern.observe(rows, True, 0.99, raw, done, scale)
        r64 = rows[:, D].cpu().double().numpy()
        vn.obs(rows[:, :D].cpu().double().numpy())
        want_r = vn.reward(r64, rows[:, D + 1].cpu().numpy() != 0)
        st = kern.get_stats()
        np.testing.assert_allclose(st["obs_mean"], vn.ob_rms.mean, rtol=1e-10)
        np.testing.assert_allclose(st["obs_var"], vn.ob_rms.var, rtol=1e-10)
        assert abs(st["obs_count"] - vn.ob_rms.count) < 1e-9 and abs(st["ret_count"] - vn.ret_rms.count) < 1e-9
        # the returns are carried in f32 (as VecNormalize's float32 array would be): 1e-6 on their statistics
        np.testing.assert_allclose(st["ret_mean"], vn.ret_rms.mean, rtol=2e-6)
        np.testing.assert_allclose(st["ret_var"], vn.ret_rms.var, rtol=2e-6)
        assert torch.equal(raw, rows[:, D]) and torch.equal(done, rows[:, D + 1])
        np.testing.assert_allclose(np.clip(raw.cpu().numpy() * scale.item(), -10, 10), want_r, rtol=2e-6)
        ret = torch.empty(n, device=DEV)
        kern.get_returns(ret)
        np.testing.assert_allclose(ret.cpu().numpy(), vn.ret, rtol=1e-5, atol=1e-4)
        assert (ret[rows[:, D + 1] != 0] == 0).all()
    assert abs(kern.get_stats()["raw_reward_sum"] - 0.0) > 1.0       # the logging sum moved


def test_gae_kernel_matches_the_oracle():
    from trex_gym import _capi
    T, n = 32, 513
    kern = _capi.Policy(n, 75, 25, 64, 0)
    g = torch.Generator(device=DEV).manual_seed(3)
    raw = -50 * torch.rand(T, n, device=DEV, generator=g)
    scale = 0.02 + 0.01 * torch.rand(T, device=DEV, generator=g)
    scale[3] = 10.0                                               # this step's rewards clip
    done = (torch.rand(T, n, device=DEV, generator=g) < 0.05).float()
    val = torch.randn(T + 1, n, device=DEV, generator=g)
    adv, ret = torch.empty(T, n, device=DEV), torch.empty(T, n, device=DEV)
    kern.gae(raw, scale, done, val, adv, ret, 0.99, 0.95, 10.0)
    rew64 = np.clip(raw.cpu().double().numpy() * scale.cpu().double().numpy()[:, None], -10, 10)
    assert (np.abs(rew64[3]) == 10).any()
    v64 = val.cpu().double().numpy()
    a64, r64 = P.gae(rew64, v64[:T], v64[T], done.cpu().double().numpy(), 0.99, 0.95)
    np.testing.assert_allclose(adv.cpu().numpy(), a64, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ret.cpu().numpy(), r64, rtol=1e-5, atol=1e-5)


def test_adam_kernel_is_tensorflows_adam_with_global_norm_clip():
    from trex_gym import _capi
    kern = _capi.Policy(64, 75, 25, 64, 0)
    Pn = kern.param_count
    g = torch.Generator(device=DEV).manual_seed(4)
    theta = torch.randn(Pn, device=DEV, generator=g)
    m, v, norm = torch.zeros(Pn, device=DEV), torch.zeros(Pn, device=DEV), torch.empty(1, device=DEV)
    p64 = {"w": theta.cpu().double().numpy().copy()}
    opt = P.Adam(p64, lr=3e-4, epsilon=1e-5)
    for it, gscale in enumerate((1e-3, 1.0, 1e-6, 0.3)):          # below and above the clip norm of 0.5; tiny gradients
        grad = gscale * torch.randn(Pn, device=DEV, generator=g)
        g64 = {"w": grad.cpu().double().numpy().copy()}
        kern.adam(theta, grad, m, v, lr=3e-4, eps=1e-5, max_grad_norm=0.5, grad_norm_out=norm)
        gc, n64 = P.clip_by_global_norm(g64, 0.5)
        p64 = opt.step(p64, gc)
        assert abs(norm.item() - n64) <= 1e-5 * n64
        assert float(grad.abs().max()) == 0.0                       # zeroed for the next accumulation
        np.testing.assert_allclose(theta.cpu().numpy(), p64["w"], rtol=0, atol=1e-5 * 3e-4 * (it + 1) + 2e-7 * np.abs(p64["w"]).max())
    kern.adam_reset()


def test_trainer_rollout_and_one_minibatch_step_match_the_oracle():
    """PPO.collect on the real env: the recorded rollout (raw rewards, reward scales, dones, values) pushed through the
    oracle's GAE gives the trainer's advantages / returns; then ONE minibatch step: loss terms and the parameter
    update against the oracle's analytic gradients + global-norm clip + TF-form Adam."""
    from trex_gym import trex_train
    from trex_gym.ppo import PPO
    env = trex_train.build_environment(256, max_episode_steps=20)
    agent = PPO(env, nsteps=24, nminibatches=4, noptepochs=1, seed=0)
    obs, act, logp0, val0, adv, ret, mean_rew = agent.collect()
    T, n = 24, 256
    assert agent.b_done.sum() > 0 and math.isfinite(mean_rew) and mean_rew < 0
    rew64 = np.clip(agent.b_rew.cpu().double().numpy() * agent.b_scale.cpu().double().numpy()[:, None], -10, 10)
    v64 = agent.b_val.cpu().double().numpy()
    a64, r64 = P.gae(rew64, v64[:T], v64[T], agent.b_done.cpu().double().numpy(), 0.99, 0.95)
    np.testing.assert_allclose(adv.cpu().numpy().reshape(T, n), a64, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ret.cpu().numpy().reshape(T, n), r64, rtol=1e-5, atol=1e-5)
    # the policy kernel's log-prob / value of the rollout == the learner's recomputation at the same parameters
    with torch.no_grad():
        d = agent.policy.dist(obs)
        torch.testing.assert_close(d.log_prob(act).sum(-1), logp0, rtol=1e-5, atol=2e-4)
        torch.testing.assert_close(agent.policy.value(obs), val0, rtol=1e-5, atol=1e-5)
    # one minibatch
    idx = torch.arange(0, T * n, 4, device=DEV)
    mb = [x.index_select(0, idx) for x in (obs, act, logp0, val0, adv, ret)]
    # (perturb the old log-probs / values so that ratios leave the clip range and the value clip bites)
    mb[2] = mb[2] + 0.3 * torch.randn_like(mb[2])
    mb[3] = mb[3] + 0.3 * torch.randn_like(mb[3])
    prm = _theta_to_params(agent.kern, agent.policy.theta)
    theta0 = agent.policy.theta.detach().cpu().double().numpy().copy()
    h = [x.cpu().double().numpy() for x in mb]
    out, grads = P.ppo_loss_and_grads(prm, h[0], h[1], -h[2], h[3], h[4], h[5], cliprange=0.2, ent_coef=0.0, vf_coef=0.5)   # (-logp = neglogp)
    loss, pg, vf, ent = agent._loss(*mb)
    assert abs(pg.item() - out["pg_loss"]) <= 1e-5 * max(1.0, abs(out["pg_loss"]))
    assert abs(vf.item() - out["vf_loss"]) <= 1e-5 * max(1.0, abs(out["vf_loss"]))
    assert abs(ent.item() - out["entropy"]) <= 1e-5 * abs(out["entropy"])
    assert ((out["ratio"] > 1.2) | (out["ratio"] < 0.8)).any()
    loss.backward()
    gflat = _grads_to_flat(agent.kern, grads)
    got = agent.policy.grad.detach().cpu().double().numpy()
    np.testing.assert_allclose(got, gflat, rtol=1e-4, atol=1e-5 * np.abs(gflat).max())
    agent.kern.adam(agent.policy.theta, agent.policy.grad, agent.adam_m, agent.adam_v, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
    gc, _ = P.clip_by_global_norm({"w": gflat}, 0.5)
    want = P.Adam({"w": theta0}, lr=3e-4, epsilon=1e-5).step({"w": theta0}, gc)["w"]
    step = agent.policy.theta.detach().cpu().double().numpy() - theta0
    # elements whose gradient is above the f32 noise of the backward pass: the update itself at 1e-5 of lr ... 2 %
    np.testing.assert_allclose(step, want - theta0, rtol=0, atol=0.02 * 3e-4)
    big = np.abs(gc["w"]) > 1e-3 * np.abs(gc["w"]).max()
    np.testing.assert_allclose(step[big], (want - theta0)[big], rtol=1e-3, atol=1e-5 * 3e-4)


@pytest.mark.parametrize("mb", [1024, 1000])
def test_native_minibatch_step_matches_the_oracle_and_autograd(mb):
    """trex_policy_minibatch_step (forward + analytic backward on the matrix cores, partial sums, clip, Adam) on a
    minibatch drawn through a permutation: the gradient against the f64 oracle's analytic gradient and against PyTorch
    autograd on the same parameters, the losses, and the parameter update against the oracle's clip + TF-form Adam.
    mb = 1000: the last tile of 32 samples is ragged."""
    from trex_gym import _capi
    from trex_gym.ppo import MlpPolicy
    torch.manual_seed(5)
    D, A, N = 75, 25, 6000
    kern = _capi.Policy(64, D, A, 64, 0)
    pol = MlpPolicy(kern.layout, kern.param_count, torch.device(DEV))
    with torch.no_grad():
        pol.theta.add_(0.05 * torch.randn_like(pol.theta))
    obs = torch.randn(N, D, device=DEV).clamp(-10, 10)
    with torch.no_grad():
        d = pol.dist(obs)
        act = d.loc + d.scale * torch.randn(N, A, device=DEV)
        logp0 = d.log_prob(act).sum(-1) + 0.3 * torch.randn(N, device=DEV)      # an older policy: ratios leave the clip range
        val0 = pol.value(obs) + 0.3 * torch.randn(N, device=DEV)
    adv = 2 * torch.randn(N, device=DEV) + 0.5
    ret = val0 + torch.randn(N, device=DEV)
    perm = torch.randperm(N, device=DEV)
    nmb = N // mb
    stats = torch.zeros(nmb, 2, device=DEV)
    kern.minibatch_stats(adv, perm, nmb, mb, stats)
    for k in range(nmb):
        a = adv[perm[k * mb:(k + 1) * mb]].double()
        assert abs(stats[k, 0].item() - a.mean().item()) < 1e-6
        assert abs(stats[k, 1].item() - 1.0 / (a.std(unbiased=False).item() + 1e-8)) < 1e-5 * stats[k, 1].item()
    first = 2 * mb
    idx = perm[first:first + mb]
    mbt = [x.index_select(0, idx) for x in (obs, act, logp0, val0, adv, ret)]
    h = [x.cpu().double().numpy() for x in mbt]
    prm = _theta_to_params(kern, pol.theta)
    theta0 = pol.theta.detach().cpu().double().numpy().copy()
    out, grads = P.ppo_loss_and_grads(prm, h[0], h[1], -h[2], h[3], h[4], h[5], cliprange=0.2, ent_coef=0.01, vf_coef=0.5)
    gflat = _grads_to_flat(kern, grads)
    assert ((out["ratio"] > 1.2) | (out["ratio"] < 0.8)).mean() > 0.1
    # autograd on the same parameters (f32)
    a_n = (mbt[4] - mbt[4].mean()) / (mbt[4].std(unbiased=False) + 1e-8)
    dd = pol.dist(mbt[0])
    ratio = (dd.log_prob(mbt[1]).sum(-1) - mbt[2]).exp()
    pg = torch.max(-a_n * ratio, -a_n * ratio.clamp(0.8, 1.2)).mean()
    v = pol.value(mbt[0])
    vclip = mbt[3] + (v - mbt[3]).clamp(-0.2, 0.2)
    vf = 0.5 * torch.max((v - mbt[5]) ** 2, (vclip - mbt[5]) ** 2).mean()
    (pg - 0.01 * dd.entropy().sum(-1).mean() + 0.5 * vf).backward()
    g_auto = pol.grad.detach().clone()
    pol.grad.zero_()
    # the native step
    m, vv, sums = torch.zeros_like(pol.theta), torch.zeros_like(pol.theta), torch.zeros(2, device=DEV)
    gbuf = torch.zeros_like(pol.theta)
    kern.minibatch_step(pol.theta, gbuf, m, vv, obs, act, logp0, val0, adv, ret, perm, first, mb, stats[2], cliprange=0.2,
                        ent_coef=0.01, vf_coef=0.5, lr=3e-4, eps=1e-5, max_grad_norm=0.5, loss_sums=sums)
    got = gbuf.cpu().double().numpy()
    scale = np.abs(gflat).max()
    np.testing.assert_allclose(got, gflat, rtol=1e-4, atol=1e-5 * scale)
    np.testing.assert_allclose(got, g_auto.cpu().double().numpy(), rtol=1e-3, atol=1e-5 * scale)
    assert abs(sums[0].item() - out["pg_loss"]) <= 1e-5 * max(1.0, abs(out["pg_loss"]))
    assert abs(sums[1].item() - out["vf_loss"]) <= 1e-5 * max(1.0, abs(out["vf_loss"]))
    gc, _ = P.clip_by_global_norm({"w": gflat}, 0.5)
    want = P.Adam({"w": theta0}, lr=3e-4, epsilon=1e-5).step({"w": theta0}, gc)["w"]
    step = pol.theta.detach().cpu().double().numpy() - theta0
    np.testing.assert_allclose(step, want - theta0, rtol=0, atol=0.02 * 3e-4)
    big = np.abs(gc["w"]) > 1e-3 * np.abs(gc["w"]).max()
    np.testing.assert_allclose(step[big], (want - theta0)[big], rtol=1e-3, atol=1e-5 * 3e-4)
    # a second step continues Adam's moments and step count (t = 2)
    kern.minibatch_step(pol.theta, gbuf, m, vv, obs, act, logp0, val0, adv, ret, perm, first, mb, stats[2], cliprange=0.2,
                        ent_coef=0.01, vf_coef=0.5, lr=3e-4, eps=1e-5, max_grad_norm=0.5, loss_sums=sums)
    assert torch.isfinite(pol.theta).all() and float((pol.theta.cpu().double() - torch.tensor(want)).abs().max()) < 2.5 * 3e-4


def test_policy_and_learner_kernels_at_other_dimensions():
    """The kernels are written for MlpPolicy's hidden = 64 but any obs_dim <= 126 / act_dim <= 32 (the double pendulum
    of tests/test_gpu_invariants.py has 6 / 2): act, GAE-free minibatch step and Adam against autograd + the oracle at
    D = 6, A = 2; unsupported shapes are refused."""
    from trex_gym import _capi
    from trex_gym.ppo import MlpPolicy
    torch.manual_seed(7)
    D, A, n = 6, 2, 200
    kern = _capi.Policy(n, D, A, 64, 0)
    pol = MlpPolicy(kern.layout, kern.param_count, torch.device(DEV))
    with torch.no_grad():
        pol.theta.add_(0.1 * torch.randn_like(pol.theta))
    rows = torch.randn(n, D + 2, device=DEV)
    kern.set_stats(dict(obs_mean=np.zeros(D), obs_var=np.ones(D), obs_count=10.0, ret_mean=0.0, ret_var=1.0, ret_count=10.0))
    noise = torch.randn(n, A, device=DEV)
    actions, obs_n, logp, val = torch.empty(n, A, device=DEV), torch.empty(n, D, device=DEV), torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    kern.act(pol.theta, rows, noise, actions, obs_n, None, logp, val)
    with torch.no_grad():
        o = (rows[:, :D] / math.sqrt(1.0 + 1e-8)).clamp(-10, 10)
        d = pol.dist(o)
        a = d.loc + d.scale * noise
        torch.testing.assert_close(actions, a, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(logp, d.log_prob(a).sum(-1), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(val, pol.value(o), rtol=1e-5, atol=1e-5)
    # one native minibatch step against autograd
    adv, ret, val0 = torch.randn(n, device=DEV), torch.randn(n, device=DEV), val + 0.2 * torch.randn(n, device=DEV)
    logp0 = logp + 0.3 * torch.randn(n, device=DEV)
    perm = torch.randperm(n, device=DEV)
    stats = torch.zeros(1, 2, device=DEV)
    kern.minibatch_stats(adv, perm, 1, n, stats)
    a_n = (adv - adv.mean()) / (adv.std(unbiased=False) + 1e-8)
    dd = pol.dist(obs_n)
    ratio = (dd.log_prob(actions).sum(-1) - logp0).exp()
    pg = torch.max(-a_n * ratio, -a_n * ratio.clamp(0.8, 1.2)).mean()
    v = pol.value(obs_n)
    vclip = val0 + (v - val0).clamp(-0.2, 0.2)
    (pg + 0.5 * 0.5 * torch.max((v - ret) ** 2, (vclip - ret) ** 2).mean()).backward()
    g_auto = pol.grad.detach().clone()
    pol.grad.zero_()
    gbuf, m, vv = torch.zeros_like(pol.theta), torch.zeros_like(pol.theta), torch.zeros_like(pol.theta)
    kern.minibatch_step(pol.theta, gbuf, m, vv, obs_n, actions, logp0, val0, adv, ret, perm, 0, n, stats[0], loss_sums=None)
    torch.testing.assert_close(gbuf, g_auto, rtol=1e-3, atol=1e-5 * float(g_auto.abs().max()))
    assert torch.isfinite(pol.theta).all()
    for bad in (dict(hidden=32), dict(act_dim=33), dict(obs_dim=127)):
        kw = dict(num_envs=8, obs_dim=6, act_dim=2, hidden=64, device=0); kw.update(bad)
        with pytest.raises(_capi.TrexError):
            _capi.Policy(**kw)
    # the learner stages 96 observation columns per sample: a wider policy can act, but not learn natively
    D2, n2 = 100, 64
    kern2 = _capi.Policy(n2, D2, A, 64, 0)
    pol2 = MlpPolicy(kern2.layout, kern2.param_count, torch.device(DEV))
    rows2 = torch.randn(n2, D2 + 2, device=DEV)
    kern2.set_stats(dict(obs_mean=np.zeros(D2), obs_var=np.ones(D2), obs_count=10.0, ret_mean=0.0, ret_var=1.0, ret_count=10.0))
    a2, o2, l2, v2 = torch.empty(n2, A, device=DEV), torch.empty(n2, D2, device=DEV), torch.empty(n2, device=DEV), torch.empty(n2, device=DEV)
    nz2 = torch.randn(n2, A, device=DEV)
    kern2.act(pol2.theta, rows2, nz2, a2, o2, None, l2, v2)
    with torch.no_grad():
        torch.testing.assert_close(v2, pol2.value((rows2[:, :D2] / math.sqrt(1.0 + 1e-8)).clamp(-10, 10)), rtol=1e-5, atol=1e-5)
    z = torch.zeros(n2, device=DEV)
    with pytest.raises(_capi.TrexError, match="obs_dim <= 96"):
        kern2.minibatch_step(pol2.theta, torch.zeros_like(pol2.theta), torch.zeros_like(pol2.theta), torch.zeros_like(pol2.theta), o2, a2,
                             l2, v2, z, z, torch.arange(n2, device=DEV), 0, n2, torch.tensor([0.0, 1.0], device=DEV), loss_sums=None)
