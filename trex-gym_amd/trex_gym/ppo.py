"""On-device PPO2 driver for TrexVecEnv (BASELINE config 3).

The reference trains with baselines' ppo2.learn on DummyVecEnv -> VecNormalize
(trex_train.py:35-63). baselines / TensorFlow are not installed here, so this is an equivalent in
plain PyTorch that keeps the WHOLE loop in HBM: observations, actions, advantages and the policy
never leave the GPU, and the env is stepped through step_tensor() (no host sync per step).

Kept from the reference (trex_train.py:49-61): MlpPolicy (2 x 64 tanh, separate value net, state-
independent log-std), lam 0.95, gamma 0.99, lr 3e-4, cliprange 0.2, ent_coef 0.0, nminibatches 32,
VecNormalize-style running normalisation of observations and of discounted returns (clip 10).
Different on purpose: nsteps (4096 steps x 4096 envs would be 16.7 M samples per batch) and
noptepochs default to smaller values; both are constructor arguments and are printed by the trainer.
"""
import math
import time

import torch
from torch import nn


class RunningMeanStd:
    """baselines.common.running_mean_std on device (parallel-variance update)."""

    def __init__(self, shape, device):
        self.mean = torch.zeros(shape, device=device, dtype=torch.float64)
        self.var = torch.ones(shape, device=device, dtype=torch.float64)
        self.count = torch.full((), 1e-4, device=device, dtype=torch.float64)   # tensor: updates are graph-capturable

    def update(self, x):
        x = x.to(torch.float64).reshape(-1, *self.mean.shape)
        bm, bv, bc = x.mean(0), x.var(0, unbiased=False), x.shape[0]
        delta = bm - self.mean
        tot = self.count + bc
        m2 = self.var * self.count + bv * bc + delta * delta * self.count * bc / tot
        self.mean.copy_(self.mean + delta * bc / tot)   # in place: the tensors keep their addresses for graph replay
        self.var.copy_(m2 / tot)
        self.count.copy_(tot)


class MlpPolicy(nn.Module):
    """baselines MlpPolicy: pi and vf are separate 2 x 64 tanh MLPs, diagonal Gaussian with a free log-std."""

    def __init__(self, obs_dim, act_dim, hidden=64):
        super().__init__()
        def mlp(out, gain):
            layers = [nn.Linear(obs_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh(), nn.Linear(hidden, out)]
            for m, g in zip([layers[0], layers[2], layers[4]], [math.sqrt(2), math.sqrt(2), gain]):
                nn.init.orthogonal_(m.weight, g)
                nn.init.zeros_(m.bias)
            return nn.Sequential(*layers)
        self.pi = mlp(act_dim, 0.01)
        self.vf = mlp(1, 1.0)
        self.logstd = nn.Parameter(torch.zeros(act_dim))

    def dist(self, obs):
        return torch.distributions.Normal(self.pi(obs), self.logstd.exp(), validate_args=False)   # the check syncs: not capturable

    def value(self, obs):
        return self.vf(obs).squeeze(-1)


class PPO:
    def __init__(self, env, nsteps=32, nminibatches=32, noptepochs=4, gamma=0.99, lam=0.95, lr=3e-4,
                 cliprange=0.2, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, clip_obs=10.0, clip_rew=10.0,
                 seed=0, use_graphs=False):
        """use_graphs: capture the whole nsteps rollout (policy + env step + normalisation) and one
        minibatch update as HIP graphs and replay them - the env step launches nothing but stream-ordered
        kernels, so it is capture-safe (tests/test_gpu_invariants.py)."""
        self.env = env
        self.dev = env.device
        self.nsteps, self.nminibatches, self.noptepochs = nsteps, nminibatches, noptepochs
        self.gamma, self.lam, self.cliprange = gamma, lam, cliprange
        self.ent_coef, self.vf_coef, self.max_grad_norm = ent_coef, vf_coef, max_grad_norm
        self.clip_obs, self.clip_rew = clip_obs, clip_rew
        torch.manual_seed(seed)
        n, od, ad = env.num_envs, env.observation_space.shape[0], env.action_space.shape[0]
        self.policy = MlpPolicy(od, ad).to(self.dev)
        self.use_graphs = use_graphs
        self.opt = torch.optim.Adam(self.policy.parameters(), lr=lr, eps=1e-5, capturable=use_graphs)
        self._rollout_graph = self._update_graph = None
        self.obs_rms = RunningMeanStd((od,), self.dev)
        self.ret_rms = RunningMeanStd((), self.dev)
        self.ret = torch.zeros(n, device=self.dev)
        self.obs = self._norm_obs(env.reset_tensor().clone(), update=True)
        self.total_env_steps = 0
        T = nsteps
        self.b_obs = torch.empty(T, n, od, device=self.dev)
        self.b_act = torch.empty(T, n, ad, device=self.dev)
        self.b_logp = torch.empty(T, n, device=self.dev)
        self.b_val = torch.empty(T + 1, n, device=self.dev)
        self.b_rew = torch.empty(T, n, device=self.dev)
        self.b_done = torch.empty(T, n, device=self.dev)
        self.b_adv = torch.empty(T, n, device=self.dev)
        self.b_ret = torch.empty(T, n, device=self.dev)
        self.raw_rew = torch.zeros((), device=self.dev, dtype=torch.float64)

    # VecNormalize (trex_train.py:45)
    def _norm_obs(self, obs, update):
        if update:
            self.obs_rms.update(obs)
        o = (obs.to(torch.float64) - self.obs_rms.mean) / torch.sqrt(self.obs_rms.var + 1e-8)
        return o.clamp(-self.clip_obs, self.clip_obs).to(torch.float32)

    def _norm_rew(self, rew, done):
        self.ret.mul_(self.gamma).add_(rew)     # in place: a captured rollout must carry the returns over replays
        self.ret_rms.update(self.ret)
        r = (rew.to(torch.float64) / torch.sqrt(self.ret_rms.var + 1e-8)).clamp(-self.clip_rew, self.clip_rew)
        self.ret.masked_fill_(done, 0.0)
        return r.to(torch.float32)

    @torch.no_grad()
    def _rollout(self):
        """nsteps env steps with the current policy + GAE, everything into the preallocated buffers."""
        T = self.nsteps
        self.raw_rew.zero_()
        for t in range(T):
            d = self.policy.dist(self.obs)
            a = d.loc + d.scale * torch.randn_like(d.loc)   # torch.normal(mean, std) checks std on the host: not capturable
            self.b_obs[t].copy_(self.obs); self.b_act[t].copy_(a)
            self.b_logp[t].copy_(d.log_prob(a).sum(-1)); self.b_val[t].copy_(self.policy.value(self.obs))
            obs, rew, done = self.env.step_tensor(a)   # the env clips to the joint limits (trex_env.py:147)
            self.raw_rew += rew.double().mean()
            self.b_rew[t].copy_(self._norm_rew(rew, done))
            self.b_done[t].copy_(done.float())
            self.obs.copy_(self._norm_obs(obs, update=True))
        self.b_val[T].copy_(self.policy.value(self.obs))
        last = torch.zeros_like(self.b_val[0])
        for t in reversed(range(T)):
            nonterm = 1.0 - self.b_done[t]
            delta = self.b_rew[t] + self.gamma * self.b_val[t + 1] * nonterm - self.b_val[t]
            last = delta + self.gamma * self.lam * nonterm * last
            self.b_adv[t].copy_(last)

    @torch.no_grad()
    def collect(self):
        T, n = self.nsteps, self.env.num_envs
        if self.use_graphs:
            if self._rollout_graph is None:
                # the GEMM library sets itself up on the first call of a shape, which a capturing stream refuses:
                # one policy evaluation outside the capture (no env step, no state change)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    self.policy.dist(self.obs); self.policy.value(self.obs)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                self._rollout_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._rollout_graph):
                    self._rollout()
            self._rollout_graph.replay()
        else:
            self._rollout()
        self.total_env_steps += T * n
        flat = lambda x: x.reshape(T * n, *x.shape[2:])
        torch.add(self.b_adv, self.b_val[:T], out=self.b_ret)   # (a persistent buffer: the captured update reads it in place)
        return (flat(self.b_obs), flat(self.b_act), flat(self.b_logp), flat(self.b_val[:T]), flat(self.b_adv), flat(self.b_ret),
                (self.raw_rew / T).item())

    def _minibatch_step(self, obs, act, logp0, val0, adv, ret):
        a = (adv - adv.mean()) / (adv.std() + 1e-8)
        d = self.policy.dist(obs)
        logp = d.log_prob(act).sum(-1)
        ratio = (logp - logp0).exp()
        pg = torch.max(-a * ratio, -a * ratio.clamp(1 - self.cliprange, 1 + self.cliprange)).mean()
        v = self.policy.value(obs)
        vclip = val0 + (v - val0).clamp(-self.cliprange, self.cliprange)
        vf = 0.5 * torch.max((v - ret) ** 2, (vclip - ret) ** 2).mean()
        ent = d.entropy().sum(-1).mean()
        loss = pg - self.ent_coef * ent + self.vf_coef * vf
        self.opt.zero_grad(set_to_none=False)
        loss.backward()
        nn.utils.clip_grad_norm_(self.policy.parameters(), self.max_grad_norm)
        self.opt.step()
        return pg.detach(), vf.detach(), ent.detach()

    def _epoch(self, srcs, N, mb):
        """One epoch: a fresh permutation, nminibatches optimiser steps; returns the summed (pg, vf, ent)."""
        perm = torch.rand(N, device=self.dev).argsort()     # (capturable: no host round trip, unlike randperm's size logic)
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
            # ONE graph per epoch (a permutation + all its minibatch steps, about 2000 nodes): replaying a graph per
            # minibatch left the update launch-bound - the 60 small kernels of a step take a third of its replay
            if self._update_graph is None:
                self._srcs = srcs              # views of the persistent rollout buffers: same addresses every iteration
                keep = [p.detach().clone() for p in self.policy.parameters()]
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):      # warm-up outside the capture (library set-up, autograd buffers)
                    for _ in range(3):
                        self._minibatch_step(*[x[:mb] for x in srcs])
                    torch.rand(N, device=self.dev).argsort()
                torch.cuda.current_stream().wait_stream(side)
                with torch.no_grad():              # undo the warm-up: same parameters and a fresh Adam state
                    for p, k in zip(self.policy.parameters(), keep):
                        p.copy_(k)
                    for st in self.opt.state.values():
                        for v in st.values():
                            if torch.is_tensor(v):
                                v.zero_()
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
