"""On-device PPO2 driver for TrexVecEnv (BASELINE config 3).

The reference trains with baselines' ppo2.learn on DummyVecEnv -> VecNormalize (trex_train.py:35-63). baselines /
TensorFlow are not installed here, so this is an equivalent that keeps the WHOLE loop in HBM: observations, actions,
advantages and the policy never leave the GPU, and the env is stepped through step_tensor() (no host sync per step).

The per-step caller work - VecNormalize's running statistics, the policy / value MLPs, the Gaussian sample and its
log-probability - runs in the hand-written HIP kernels of include/trex_policy.h (csrc/policy_step.hip: two launches
per env step instead of ~30 framework kernels), GAE(lambda) and the optimiser step (global-norm clip + Adam in
TensorFlow's form, as ppo2 applies it) in one launch each. There is no fallback: without libtrex_hip.so this module
does not import. PyTorch keeps one job: autograd over the minibatch loss, on parameters that are VIEWS into the flat
f32 vector the kernels read (weights stored [in, out]).

Kept from the reference (trex_train.py:49-61): MlpPolicy (2 x 64 tanh, separate value net, state-independent
log-std), lam 0.95, gamma 0.99, lr 3e-4, cliprange 0.2, ent_coef 0.0, nminibatches 32, VecNormalize's running
normalisation of observations and of discounted returns (clip 10). Different on purpose: nsteps (4096 steps x 4096
envs would be 16.7 M samples per batch) and noptepochs default to smaller values; both are constructor arguments and
are printed by the trainer. The arithmetic is restated in f64 by oracle/ppo_oracle.py (tests/test_gpu_policy.py).
"""
import math
import time

import torch
from torch import nn

from . import _capi


class MlpPolicy(nn.Module):
    """baselines MlpPolicy: pi and vf are separate 2 x 64 tanh MLPs, diagonal Gaussian with a free log-std. The
    parameters are views into ONE flat f32 vector laid out as include/trex_policy.h says (weights [in, out]); their
    gradients are views into a second flat vector, so that the optimiser step is one kernel over both."""

    def __init__(self, layout, param_count, device):
        super().__init__()
        self.theta = torch.zeros(param_count, device=device)
        self.grad = torch.zeros(param_count, device=device)
        self._names = list(layout)
        for name, (off, shape) in layout.items():
            n = math.prod(shape)
            p = nn.Parameter(self.theta[off:off + n].view(shape))
            p.grad = self.grad[off:off + n].view(shape)      # autograd accumulates in place: the views stay
            self.register_parameter(name.replace(".", "_"), p)
        with torch.no_grad():
            for net, gain_out in (("pi", 0.01), ("vf", 1.0)):
                for k, gain in ((1, math.sqrt(2)), (2, math.sqrt(2)), (3, gain_out)):
                    w = self.p("%s.W%d" % (net, k))
                    w.copy_(nn.init.orthogonal_(torch.empty(w.shape[1], w.shape[0], device=device), gain).t())

    def p(self, name):
        return getattr(self, name.replace(".", "_"))

    def _mlp(self, net, x):
        h = torch.tanh(torch.addmm(self.p(net + ".b1"), x, self.p(net + ".W1")))
        h = torch.tanh(torch.addmm(self.p(net + ".b2"), h, self.p(net + ".W2")))
        return torch.addmm(self.p(net + ".b3"), h, self.p(net + ".W3"))

    def mean(self, obs):
        return self._mlp("pi", obs)

    def dist(self, obs):
        return torch.distributions.Normal(self.mean(obs), self.logstd.exp(), validate_args=False)   # the check syncs: not capturable

    def value(self, obs):
        return self._mlp("vf", obs).squeeze(-1)


def merge_running_moments(prefix, parts):
    """VecNormalize's running statistics across data-parallel ranks. All ranks start a rollout with the SAME statistics
    `prefix` = (mean, var, count); during the rollout rank r's statistics kernel folds in its own envs' observations and
    ends with parts[r] = (mean_r, var_r, count_r). The statistics of everything seen - the prefix and every rank's new
    data once - by Chan's parallel formula, in rank order (f64; the same on every rank):
      new data of rank r: n_r = count_r - c0, mean and M2 by un-merging the prefix from rank r's statistics;
      merged = prefix (+) new_0 (+) new_1 ...     with (a (+) b): delta = mean_b - mean_a, n = n_a + n_b,
               mean = mean_a + delta n_b / n, M2 = M2_a + M2_b + delta^2 n_a n_b / n."""
    import numpy as np
    m0, v0, c0 = (np.asarray(x, np.float64) for x in prefix)
    mean, M2, cnt = m0.copy(), v0 * c0, float(c0)
    for mr, vr, cr in parts:
        mr, vr, cr = np.asarray(mr, np.float64), np.asarray(vr, np.float64), float(cr)
        nb = cr - float(c0)
        if nb <= 0:
            continue
        mb_ = (cr * mr - c0 * m0) / nb                                   # mean of rank r's new data
        M2b = vr * cr - v0 * c0 - (mb_ - m0) ** 2 * c0 * nb / cr         # its sum of squared deviations
        delta = mb_ - mean
        tot = cnt + nb
        M2 = M2 + M2b + delta ** 2 * cnt * nb / tot
        mean = mean + delta * nb / tot
        cnt = tot
    return mean, M2 / cnt, cnt


def dp_minibatch_stats(adv, perm, nminibatches, mb, world, out, group=None):
    """mean and 1 / (std + 1e-8) (population std) of the advantages of every GLOBAL minibatch - this rank's share
    perm[k mb : (k + 1) mb) together with the other ranks' - from f64 sums, ONE all-reduce per epoch -> out [nminibatches, 2]."""
    import torch.distributed as dist
    a = adv[perm[: nminibatches * mb]].view(nminibatches, mb).double()
    s = torch.stack([a.sum(1), (a * a).sum(1)], 1)
    dist.all_reduce(s, group=group)
    n = float(mb * world)
    mean = s[:, 0] / n
    var = (s[:, 1] / n - mean * mean).clamp_min(0.0)
    out.copy_(torch.stack([mean, 1.0 / (var.sqrt() + 1e-8)], 1).float())


def dp_minibatch_step(kern, theta, grad, m, v, srcs, perm, first, mb, stats_row, world, group=None, cliprange=0.2, ent_coef=0.0,
                      vf_coef=0.5, lr=3e-4, max_grad_norm=0.5, loss_sums=None):
    """One data-parallel minibatch step: this rank's gradient / ranks (trex_policy_minibatch_grad), summed over the ranks (the
    one collective of the trainer: 21 112 floats), then global-norm clip + TF-form Adam on every rank (trex_policy_adam) - the
    update of the concatenated minibatch, the same on every rank."""
    import torch.distributed as dist
    obs, act, logp0, val0, adv, ret = srcs
    kern.minibatch_grad(theta, grad, obs, act, logp0, val0, adv, ret, perm, first, mb, stats_row, cliprange, ent_coef, vf_coef,
                        1.0 / world, loss_sums)
    dist.all_reduce(grad, group=group)
    kern.adam(theta, grad, m, v, lr=lr, eps=1e-5, max_grad_norm=max_grad_norm)


class PPO:
    def __init__(self, env, nsteps=32, nminibatches=32, noptepochs=4, gamma=0.99, lam=0.95, lr=3e-4,
                 cliprange=0.2, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, clip_obs=10.0, clip_rew=10.0,
                 seed=0, use_graphs=False, native_learner=True):
        """native_learner: the minibatch step is three HIP launches (trex_policy_minibatch_step: forward + analytic
        backward on the matrix cores, then ordered partial sums, then clip + Adam) instead of ~90 autograd kernels; False keeps
        the PyTorch autograd path (the f32 reference the native step is tested against).
        use_graphs: capture the whole nsteps rollout (policy kernel + env step + statistics kernel per step) and
        one epoch of minibatch updates as HIP graphs and replay them - everything launched is stream-ordered, so it
        is capture-safe (tests/test_gpu_invariants.py)."""
        self.env = env
        self.dev = env.device
        # DATA-PARALLEL (BASELINE config 3 on N GPUs; SURVEY 8e: "only PPO gradient all-reduce remains"): one process per GPU, env
        # sharded as the env itself is (TrexVecEnv(rank=, world_size=)): every rank rolls out its own envs with its own replica of
        # the policy; per minibatch step the ranks SUM their gradients (one all-reduce of the flat 21 112-float vector between the
        # reduce launch and the Adam launch), per epoch the advantage statistics of the minibatches, per rollout the VecNormalize
        # moments (Chan's formula) - so that every rank applies the update of the concatenated minibatch and keeps identical
        # parameters and statistics. Unmeasured on more than one GPU.
        self.rank, self.world = int(getattr(env, "rank", 0)), int(getattr(env, "world_size", 1))
        self.group = getattr(env, "process_group", None)
        if self.world > 1 and (use_graphs or not native_learner):
            raise ValueError("the data-parallel trainer runs the native learner, eagerly (collectives inside a captured graph are not supported here)")
        self.nsteps, self.nminibatches, self.noptepochs = nsteps, nminibatches, noptepochs
        self.gamma, self.lam, self.cliprange, self.lr = gamma, lam, cliprange, lr
        self.ent_coef, self.vf_coef, self.max_grad_norm = ent_coef, vf_coef, max_grad_norm
        self.clip_obs, self.clip_rew = clip_obs, clip_rew
        torch.manual_seed(seed)
        n, od, ad = env.num_envs, env.observation_space.shape[0], env.action_space.shape[0]
        self.kern = _capi.Policy(n, od, ad, 64, self.dev.index)       # VecNormalize state + the kernels
        self.policy = MlpPolicy(self.kern.layout, self.kern.param_count, self.dev)
        self.adam_m = torch.zeros_like(self.policy.theta)
        self.adam_v = torch.zeros_like(self.policy.theta)
        self.use_graphs, self.native_learner = use_graphs, native_learner
        self._rollout_graph = self._update_graph = None
        self.mb_stats = torch.zeros(nminibatches, 2, device=self.dev)
        self.loss_sums = torch.zeros(2, device=self.dev)
        self.total_env_steps = 0
        if self.world > 1:      # same initial parameters everywhere (seeded above), different exploration noise per rank
            import torch.distributed as dist
            dist.broadcast(self.policy.theta.data, 0, group=self.group)
            torch.manual_seed(seed + 1000003 * (self.rank + 1))
        T = nsteps
        self.noise = torch.empty(T, n, ad, device=self.dev)
        self.actions = torch.empty(n, ad, device=self.dev)
        self.b_obs = torch.empty(T, n, od, device=self.dev)
        self.b_act = torch.empty(T, n, ad, device=self.dev)
        self.b_logp = torch.empty(T, n, device=self.dev)
        self.b_val = torch.empty(T + 1, n, device=self.dev)
        self.b_rew = torch.empty(T, n, device=self.dev)          # RAW rewards; GAE applies scale[t] and the clip
        self.b_scale = torch.empty(T, device=self.dev)
        self.b_done = torch.empty(T, n, device=self.dev)
        self.b_adv = torch.empty(T, n, device=self.dev)
        self.b_ret = torch.empty(T, n, device=self.dev)
        self._raw_sum = 0.0
        env.reset_tensor()
        self.kern.observe(env.rows, with_reward=False)           # VecNormalize.reset: the statistics see the first obs

    @property
    def obs(self):
        """The normalised observation the policy saw at the last step of the last rollout (diagnostics)."""
        return self.b_obs[-1]

    @torch.no_grad()
    def _rollout(self):
        """nsteps env steps with the current policy + GAE, everything into the preallocated buffers. Per step: ONE
        policy launch (normalise, both MLPs on the matrix cores, sample, log-prob), the env's step launch, ONE
        statistics launch (VecNormalize)."""
        T, env, k, th = self.nsteps, self.env, self.kern, self.policy.theta
        self.noise.normal_()
        for t in range(T):
            k.act(th, env.rows, self.noise[t], self.actions, self.b_obs[t], self.b_act[t], self.b_logp[t], self.b_val[t],
                  clip_obs=self.clip_obs)
            env.step_tensor(self.actions)       # the env clips to the joint limits (trex_env.py:147)
            k.observe(env.rows, True, self.gamma, self.b_rew[t], self.b_done[t], self.b_scale[t:t + 1])
        k.act(th, env.rows, None, None, value_out=self.b_val[T], clip_obs=self.clip_obs, value_only=True)
        k.gae(self.b_rew, self.b_scale, self.b_done, self.b_val, self.b_adv, self.b_ret, self.gamma, self.lam, self.clip_rew)

    @torch.no_grad()
    def collect(self):
        T, n = self.nsteps, self.env.num_envs
        if len(getattr(self.env, "_row_blocks", [0])) != 1:
            raise ValueError("the trainer reads env.rows in place: TrexVecEnv(row_buffers=1)")
        if self.use_graphs:
            if self._rollout_graph is None:
                torch.cuda.synchronize()
                self._rollout_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._rollout_graph):
                    self._rollout()
            self._rollout_graph.replay()
        else:
            self._rollout()
        self.total_env_steps += T * n * self.world
        if self.world > 1:
            self._merge_statistics()
        flat = lambda x: x.reshape(T * n, *x.shape[2:])
        raw = self.kern.get_stats()["raw_reward_sum"]            # (the one host read-back per rollout)
        mean_rew, self._raw_sum = (raw - self._raw_sum) / (T * n), raw
        return (flat(self.b_obs), flat(self.b_act), flat(self.b_logp), flat(self.b_val[:T]), flat(self.b_adv), flat(self.b_ret),
                mean_rew)

    def _merge_statistics(self):
        """Once per rollout: every rank's VecNormalize moments (observations, returns) merged into the same statistics."""
        import numpy as np
        import torch.distributed as dist
        st = self.kern.get_stats()
        prev = getattr(self, "_stats_prefix", None)
        mine = {k: st[k] for k in ("obs_mean", "obs_var", "obs_count", "ret_mean", "ret_var", "ret_count") if k in st}
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=self.group)
        if prev is None:      # first rollout: the statistics started from the reset's first observation on every rank
            prev = dict(obs_mean=np.zeros_like(np.asarray(st["obs_mean"], np.float64)), obs_var=np.ones_like(np.asarray(st["obs_var"], np.float64)),
                        obs_count=1e-4, ret_mean=0.0, ret_var=1.0, ret_count=1e-4)
        om, ov, oc = merge_running_moments((prev["obs_mean"], prev["obs_var"], prev["obs_count"]),
                                           [(e["obs_mean"], e["obs_var"], e["obs_count"]) for e in everyone])
        st.update(obs_mean=om, obs_var=ov, obs_count=oc)
        if "ret_count" in mine:
            rm, rv, rc = merge_running_moments((prev["ret_mean"], prev["ret_var"], prev["ret_count"]),
                                               [(e["ret_mean"], e["ret_var"], e["ret_count"]) for e in everyone])
            st.update(ret_mean=float(rm), ret_var=float(rv), ret_count=rc)
        self.kern.set_stats(st)
        self._stats_prefix = {k: (np.array(st[k], np.float64) if hasattr(st[k], "__len__") else float(st[k])) for k in mine}

    def _dp_minibatch_stats(self, adv, perm, mb):
        dp_minibatch_stats(adv, perm, self.nminibatches, mb, self.world, self.mb_stats, self.group)

    def dp_minibatch_step(self, srcs, perm, first, mb, stats_row):
        dp_minibatch_step(self.kern, self.policy.theta, self.policy.grad, self.adam_m, self.adam_v, srcs, perm, first, mb, stats_row,
                          self.world, self.group, self.cliprange, self.ent_coef, self.vf_coef, self.lr, self.max_grad_norm, self.loss_sums)

    def _loss(self, obs, act, logp0, val0, adv, ret):
        a = (adv - adv.mean()) / (adv.std(unbiased=False) + 1e-8)       # numpy's std: population
        d = self.policy.dist(obs)
        logp = d.log_prob(act).sum(-1)
        ratio = (logp - logp0).exp()
        pg = torch.max(-a * ratio, -a * ratio.clamp(1 - self.cliprange, 1 + self.cliprange)).mean()
        v = self.policy.value(obs)
        vclip = val0 + (v - val0).clamp(-self.cliprange, self.cliprange)
        vf = 0.5 * torch.max((v - ret) ** 2, (vclip - ret) ** 2).mean()
        ent = d.entropy().sum(-1).mean()
        return pg - self.ent_coef * ent + self.vf_coef * vf, pg, vf, ent

    def _minibatch_step(self, obs, act, logp0, val0, adv, ret):
        loss, pg, vf, ent = self._loss(obs, act, logp0, val0, adv, ret)
        loss.backward()          # accumulates into the flat gradient vector (zeroed by the optimiser kernel)
        self.kern.adam(self.policy.theta, self.policy.grad, self.adam_m, self.adam_v, lr=self.lr, eps=1e-5,
                       max_grad_norm=self.max_grad_norm)
        return pg.detach(), vf.detach(), ent.detach()

    def _epoch(self, srcs, N, mb):
        """One epoch: a fresh permutation, nminibatches optimiser steps; returns the summed (pg, vf, ent)."""
        perm = torch.rand(N, device=self.dev).argsort()     # (capturable: no host round trip, unlike randperm's size logic)
        if self.native_learner:
            obs, act, logp0, val0, adv, ret = srcs
            k, pol = self.kern, self.policy
            self.loss_sums.zero_()
            if self.world > 1:
                self._dp_minibatch_stats(adv, perm, mb)
                for i in range(self.nminibatches):
                    self.dp_minibatch_step(srcs, perm, i * mb, mb, self.mb_stats[i])
                import torch.distributed as dist
                dist.all_reduce(self.loss_sums, group=self.group)
                ent = (pol.logstd.detach() + 0.5 * (math.log(2 * math.pi) + 1.0)).sum() * self.nminibatches
                return torch.cat([self.loss_sums, ent.reshape(1)])
            k.minibatch_stats(adv, perm, self.nminibatches, mb, self.mb_stats)
            for i in range(self.nminibatches):
                k.minibatch_step(pol.theta, pol.grad, self.adam_m, self.adam_v, obs, act, logp0, val0, adv, ret, perm, i * mb, mb,
                                 self.mb_stats[i], self.cliprange, self.ent_coef, self.vf_coef, self.lr, 0.9, 0.999, 1e-5,
                                 self.max_grad_norm, self.loss_sums)
            ent = (pol.logstd.detach() + 0.5 * (math.log(2 * math.pi) + 1.0)).sum() * self.nminibatches   # (of the updated policy)
            return torch.cat([self.loss_sums, ent.reshape(1)])
        acc = torch.zeros(3, device=self.dev)
        for k in range(self.nminibatches):
            idx = perm[k * mb:(k + 1) * mb]
            out = self._minibatch_step(*[x.index_select(0, idx) for x in srcs])
            acc = acc + torch.stack(out)
        return acc

    def update(self, batch):
        obs, act, logp0, val0, adv, ret, _ = batch
        srcs = (obs, act, logp0, val0, adv, ret)
        N = obs.shape[0]
        mb = N // self.nminibatches
        if self.use_graphs:
            # ONE graph per epoch (a permutation + all its minibatch steps)
            if self._update_graph is None:
                self._srcs = srcs              # views of the persistent rollout buffers: same addresses every iteration
                keep = self.policy.theta.clone()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):      # warm-up outside the capture (library set-up, autograd buffers, workspace)
                    if self.native_learner:
                        self._epoch(srcs, N, mb)
                    else:
                        for _ in range(3):
                            self._minibatch_step(*[x[:mb] for x in srcs])
                    torch.rand(N, device=self.dev).argsort()
                torch.cuda.current_stream().wait_stream(side)
                with torch.no_grad():              # undo the warm-up: same parameters and a fresh Adam state
                    self.policy.theta.copy_(keep)
                    self.adam_m.zero_(); self.adam_v.zero_(); self.policy.grad.zero_()
                    self.kern.adam_reset()
                torch.cuda.synchronize()
                self._update_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._update_graph):
                    self._epoch_out = self._epoch(srcs, N, mb)
            assert all(a.data_ptr() == b.data_ptr() for a, b in zip(srcs, self._srcs)), "rollout buffers moved"
            total = torch.zeros(3, device=self.dev)
            for _ in range(self.noptepochs):
                self._update_graph.replay()
                total += self._epoch_out
        else:
            total = torch.zeros(3, device=self.dev)
            for _ in range(self.noptepochs):
                total += self._epoch(srcs, N, mb)
        pg, vf, ent = (total / (self.noptepochs * self.nminibatches)).tolist()
        return dict(policy_loss=pg, value_loss=vf, entropy=ent)

    def learn(self, total_timesteps, log=print):
        t0 = time.perf_counter()
        it = 0
        history = []
        while self.total_env_steps < total_timesteps:
            batch = self.collect()
            info = self.update(batch)
            it += 1
            info.update(iteration=it, env_steps=self.total_env_steps, mean_step_reward=batch[-1],
                        env_steps_per_s=self.total_env_steps / (time.perf_counter() - t0))
            history.append(info)
            if log:
                log("it %3d  env-steps %9d  mean reward/step %12.4f  pg %.4f  vf %.4f  ent %.3f  %.0f env-steps/s"
                    % (it, info["env_steps"], info["mean_step_reward"], info["policy_loss"], info["value_loss"],
                       info["entropy"], info["env_steps_per_s"]))
        return history
