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
        self.count = 1e-4

    def update(self, x):
        x = x.to(torch.float64).reshape(-1, *self.mean.shape)
        bm, bv, bc = x.mean(0), x.var(0, unbiased=False), x.shape[0]
        delta = bm - self.mean
        tot = self.count + bc
        self.mean = self.mean + delta * bc / tot
        m2 = self.var * self.count + bv * bc + delta * delta * self.count * bc / tot
        self.var = m2 / tot
        self.count = tot


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
        return torch.distributions.Normal(self.pi(obs), self.logstd.exp())

    def value(self, obs):
        return self.vf(obs).squeeze(-1)


class PPO:
    def __init__(self, env, nsteps=32, nminibatches=32, noptepochs=4, gamma=0.99, lam=0.95, lr=3e-4,
                 cliprange=0.2, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, clip_obs=10.0, clip_rew=10.0,
                 seed=0):
        self.env = env
        self.dev = env.device
        self.nsteps, self.nminibatches, self.noptepochs = nsteps, nminibatches, noptepochs
        self.gamma, self.lam, self.cliprange = gamma, lam, cliprange
        self.ent_coef, self.vf_coef, self.max_grad_norm = ent_coef, vf_coef, max_grad_norm
        self.clip_obs, self.clip_rew = clip_obs, clip_rew
        torch.manual_seed(seed)
        n, od, ad = env.num_envs, env.observation_space.shape[0], env.action_space.shape[0]
        self.policy = MlpPolicy(od, ad).to(self.dev)
        self.opt = torch.optim.Adam(self.policy.parameters(), lr=lr, eps=1e-5)
        self.obs_rms = RunningMeanStd((od,), self.dev)
        self.ret_rms = RunningMeanStd((), self.dev)
        self.ret = torch.zeros(n, device=self.dev)
        self.obs = self._norm_obs(env.reset_tensor().clone(), update=True)
        self.total_env_steps = 0

    # VecNormalize (trex_train.py:45)
    def _norm_obs(self, obs, update):
        if update:
            self.obs_rms.update(obs)
        o = (obs.to(torch.float64) - self.obs_rms.mean) / torch.sqrt(self.obs_rms.var + 1e-8)
        return o.clamp(-self.clip_obs, self.clip_obs).to(torch.float32)

    def _norm_rew(self, rew, done):
        self.ret = self.ret * self.gamma + rew
        self.ret_rms.update(self.ret)
        r = (rew.to(torch.float64) / torch.sqrt(self.ret_rms.var + 1e-8)).clamp(-self.clip_rew, self.clip_rew)
        self.ret = torch.where(done, torch.zeros_like(self.ret), self.ret)
        return r.to(torch.float32)

    @torch.no_grad()
    def collect(self):
        T, n = self.nsteps, self.env.num_envs
        od, ad = self.obs.shape[1], self.env.action_space.shape[0]
        b_obs = torch.empty(T, n, od, device=self.dev)
        b_act = torch.empty(T, n, ad, device=self.dev)
        b_logp = torch.empty(T, n, device=self.dev)
        b_val = torch.empty(T + 1, n, device=self.dev)
        b_rew = torch.empty(T, n, device=self.dev)
        b_done = torch.empty(T, n, device=self.dev)
        raw_rew = torch.zeros((), device=self.dev, dtype=torch.float64)
        for t in range(T):
            d = self.policy.dist(self.obs)
            a = d.sample()
            b_obs[t], b_act[t], b_logp[t], b_val[t] = self.obs, a, d.log_prob(a).sum(-1), self.policy.value(self.obs)
            obs, rew, done = self.env.step_tensor(a)   # the env clips to the joint limits (trex_env.py:147)
            raw_rew += rew.double().mean()
            b_rew[t] = self._norm_rew(rew, done)
            b_done[t] = done.float()
            self.obs = self._norm_obs(obs, update=True)
        b_val[T] = self.policy.value(self.obs)
        adv = torch.empty(T, n, device=self.dev)
        last = torch.zeros(n, device=self.dev)
        for t in reversed(range(T)):
            nonterm = 1.0 - b_done[t]
            delta = b_rew[t] + self.gamma * b_val[t + 1] * nonterm - b_val[t]
            last = delta + self.gamma * self.lam * nonterm * last
            adv[t] = last
        ret = adv + b_val[:T]
        self.total_env_steps += T * n
        flat = lambda x: x.reshape(T * n, *x.shape[2:])
        return flat(b_obs), flat(b_act), flat(b_logp), flat(b_val[:T]), flat(adv), flat(ret), (raw_rew / T).item()

    def update(self, batch):
        obs, act, logp0, val0, adv, ret, _ = batch
        N = obs.shape[0]
        mb = N // self.nminibatches
        stats = []
        for _ in range(self.noptepochs):
            perm = torch.randperm(N, device=self.dev)
            for k in range(self.nminibatches):
                idx = perm[k * mb:(k + 1) * mb]
                a = adv[idx]
                a = (a - a.mean()) / (a.std() + 1e-8)
                d = self.policy.dist(obs[idx])
                logp = d.log_prob(act[idx]).sum(-1)
                ratio = (logp - logp0[idx]).exp()
                pg = torch.max(-a * ratio, -a * ratio.clamp(1 - self.cliprange, 1 + self.cliprange)).mean()
                v = self.policy.value(obs[idx])
                vclip = val0[idx] + (v - val0[idx]).clamp(-self.cliprange, self.cliprange)
                vf = 0.5 * torch.max((v - ret[idx]) ** 2, (vclip - ret[idx]) ** 2).mean()
                ent = d.entropy().sum(-1).mean()
                loss = pg - self.ent_coef * ent + self.vf_coef * vf
                self.opt.zero_grad(set_to_none=True)
                loss.backward()
                nn.utils.clip_grad_norm_(self.policy.parameters(), self.max_grad_norm)
                self.opt.step()
            stats.append((pg.detach(), vf.detach(), ent.detach()))
        pg, vf, ent = (torch.stack(x).mean().item() for x in zip(*stats))
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
