"""CPU restatement (numpy, f64) of the trainer-side arithmetic of the reference's training script:
baselines' ppo2.learn on DummyVecEnv -> VecNormalize with MlpPolicy (trex_train.py:41-61).

TEST INFRASTRUCTURE: only tests/ may import this module; the product (trex_gym/ppo.py and its HIP policy
kernel) never does.

PARITY UNPINNED: `baselines` (OpenAI baselines, un-pinned in the reference: setup.py:12, README.md:42-53; the
call signature at trex_train.py:49-61 dates it to mid-2018) and TensorFlow 1.x are third-party dependencies
that are neither vendored in /root/reference nor installed here, and the reference holds no fixture of any
trainer quantity. What follows restates the PUBLISHED algorithms they implement:

  * PPO, clipped surrogate objective: Schulman et al. 2017, "Proximal Policy Optimization Algorithms", eq. (7),
    with the value-function clipping and per-minibatch advantage normalisation of the ppo2 implementation;
  * GAE(lambda): Schulman et al. 2016, "High-Dimensional Continuous Control Using Generalized Advantage
    Estimation", eq. (16), truncated at the rollout end with a bootstrap value and cut at episode ends;
  * Adam: Kingma & Ba 2015, in the form TensorFlow 1.x documents for tf.train.AdamOptimizer (the "epsilon hat"
    form: lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); theta -= lr_t m / (sqrt(v) + eps)), eps = 1e-5 as ppo2 passes it;
  * running mean / variance: Chan, Golub & LeVeque's parallel update (what VecNormalize's RunningMeanStd uses),
    count initialised to 1e-4, mean 0, variance 1;
  * MlpPolicy: two separate 2 x 64 tanh MLPs (policy mean, value), state-independent log-std, diagonal Gaussian.

Hyper-parameters from the reference's own call (trex_train.py:49-61): lam 0.95, gamma 0.99, lr 3e-4, cliprange 0.2,
ent_coef 0.0, nminibatches 32, noptepochs 32; ppo2's defaults for what the call leaves out: vf_coef 0.5,
max_grad_norm 0.5; VecNormalize's defaults (trex_train.py:45 passes none): clipob 10, cliprew 10, gamma 0.99,
epsilon 1e-8.
"""
import numpy as np

LOG_2PI = np.log(2.0 * np.pi)


class RunningMeanStd:
    """Parallel-variance running statistics (Chan et al.); VecNormalize keeps one for observations, one for returns."""

    def __init__(self, shape=(), epsilon=1e-4):
        self.mean = np.zeros(shape, np.float64)
        self.var = np.ones(shape, np.float64)
        self.count = float(epsilon)

    def update(self, x):
        x = np.asarray(x, np.float64).reshape((-1,) + self.mean.shape)
        bm, bv, bc = x.mean(0), x.var(0), x.shape[0]      # population variance of the batch
        delta = bm - self.mean
        tot = self.count + bc
        m2 = self.var * self.count + bv * bc + delta * delta * self.count * bc / tot
        self.mean = self.mean + delta * bc / tot
        self.var = m2 / tot
        self.count = tot


class VecNormalize:
    """Observation / reward normalisation as the wrapper at trex_train.py:45 applies it to every env step."""

    def __init__(self, num_envs, obs_dim, gamma=0.99, clipob=10.0, cliprew=10.0, epsilon=1e-8):
        self.ob_rms, self.ret_rms = RunningMeanStd((obs_dim,)), RunningMeanStd(())
        self.ret = np.zeros(num_envs)
        self.gamma, self.clipob, self.cliprew, self.epsilon = gamma, clipob, cliprew, epsilon

    def obs(self, obs, update=True):
        if update:
            self.ob_rms.update(obs)
        return np.clip((obs - self.ob_rms.mean) / np.sqrt(self.ob_rms.var + self.epsilon), -self.clipob, self.clipob)

    def reward(self, rews, news):
        self.ret = self.ret * self.gamma + rews
        self.ret_rms.update(self.ret)
        out = np.clip(rews / np.sqrt(self.ret_rms.var + self.epsilon), -self.cliprew, self.cliprew)
        self.ret[np.asarray(news, bool)] = 0.0
        return out


def mlp_forward(params, prefix, x):
    """2 x 64 tanh MLP; params[prefix + '.0.weight'] etc. in torch's nn.Sequential naming ([out, in] weights).
    Returns (output, activations needed by the backward pass)."""
    h1 = np.tanh(x @ params[prefix + ".0.weight"].T + params[prefix + ".0.bias"])
    h2 = np.tanh(h1 @ params[prefix + ".2.weight"].T + params[prefix + ".2.bias"])
    out = h2 @ params[prefix + ".4.weight"].T + params[prefix + ".4.bias"]
    return out, (x, h1, h2)


def mlp_backward(params, prefix, acts, dout, grads):
    x, h1, h2 = acts
    grads[prefix + ".4.weight"] = dout.T @ h2
    grads[prefix + ".4.bias"] = dout.sum(0)
    d2 = (dout @ params[prefix + ".4.weight"]) * (1.0 - h2 * h2)
    grads[prefix + ".2.weight"] = d2.T @ h1
    grads[prefix + ".2.bias"] = d2.sum(0)
    d1 = (d2 @ params[prefix + ".2.weight"]) * (1.0 - h1 * h1)
    grads[prefix + ".0.weight"] = d1.T @ x
    grads[prefix + ".0.bias"] = d1.sum(0)


def policy_forward(params, obs):
    """mean [n, A], logstd [A], value [n]"""
    mean, _ = mlp_forward(params, "pi", obs)
    value, _ = mlp_forward(params, "vf", obs)
    return mean, params["logstd"], value[:, 0]


def neglogp(mean, logstd, act):
    """-log N(act; mean, exp(logstd)) of a diagonal Gaussian, summed over the action dimensions."""
    z = (act - mean) / np.exp(logstd)
    return 0.5 * np.sum(z * z, -1) + 0.5 * LOG_2PI * act.shape[-1] + np.sum(logstd)


def sample_action(mean, logstd, noise):
    """act = mean + std * noise (noise ~ N(0, 1) supplied by the caller, so that both sides use the same draws)."""
    return mean + np.exp(logstd) * noise


def gae(rewards, values, last_values, dones, gamma=0.99, lam=0.95):
    """rewards, values, dones [T, n] (dones[t] = the flag RETURNED by step t), last_values [n] = V(obs after step T-1).
    Returns (advantages, returns) [T, n]."""
    T = rewards.shape[0]
    adv = np.zeros_like(rewards, dtype=np.float64)
    last = np.zeros(rewards.shape[1])
    for t in reversed(range(T)):
        nonterm = 1.0 - dones[t]
        nextv = last_values if t == T - 1 else values[t + 1]
        delta = rewards[t] + gamma * nextv * nonterm - values[t]
        last = delta + gamma * lam * nonterm * last
        adv[t] = last
    return adv, adv + values


def ppo_loss_and_grads(params, obs, act, neglogp_old, value_old, adv, ret, cliprange=0.2, ent_coef=0.0, vf_coef=0.5):
    """One minibatch: advantage normalisation (population std + 1e-8), clipped surrogate, clipped value loss, entropy
    bonus. Returns (dict of scalars, dict of gradients of the total loss w.r.t. every parameter)."""
    n, A = act.shape
    a = (adv - adv.mean()) / (adv.std() + 1e-8)
    mean, acts_pi = mlp_forward(params, "pi", obs)
    vout, acts_vf = mlp_forward(params, "vf", obs)
    v = vout[:, 0]
    logstd = params["logstd"]
    std = np.exp(logstd)
    z = (act - mean) / std
    nlp = 0.5 * np.sum(z * z, -1) + 0.5 * LOG_2PI * A + np.sum(logstd)
    ratio = np.exp(neglogp_old - nlp)
    l1, l2 = -a * ratio, -a * np.clip(ratio, 1.0 - cliprange, 1.0 + cliprange)
    pg_loss = np.mean(np.maximum(l1, l2))
    vclip = value_old + np.clip(v - value_old, -cliprange, cliprange)
    e1, e2 = (v - ret) ** 2, (vclip - ret) ** 2
    vf_loss = 0.5 * np.mean(np.maximum(e1, e2))
    entropy = np.sum(logstd + 0.5 * (LOG_2PI + 1.0))
    loss = pg_loss - ent_coef * entropy + vf_coef * vf_loss
    # ---- gradients
    grads = {}
    # d pg / d ratio: the unclipped branch is active where l1 >= l2 (max picks it; ties -> l1, whose gradient is
    # -a: where the clip is inactive both branches agree anyway)
    unclipped = l1 >= l2
    inside = (ratio >= 1.0 - cliprange) & (ratio <= 1.0 + cliprange)
    dratio = np.where(unclipped, -a, np.where(inside, -a, 0.0)) / n
    dnlp = -dratio * ratio                                  # ratio = exp(old - nlp)
    dmean = dnlp[:, None] * (-(z / std))                    # d nlp / d mean = -(act - mean) / std^2
    dlogstd = np.sum(dnlp[:, None] * (1.0 - z * z), 0) - ent_coef * np.ones(A)
    mlp_backward(params, "pi", acts_pi, dmean, grads)
    grads["logstd"] = dlogstd
    dv_un = 2.0 * (v - ret)
    dv_cl = 2.0 * (vclip - ret) * ((v - value_old >= -cliprange) & (v - value_old <= cliprange))
    dv = vf_coef * 0.5 * np.where(e1 >= e2, dv_un, dv_cl) / n
    mlp_backward(params, "vf", acts_vf, dv[:, None], grads)
    return dict(loss=loss, pg_loss=pg_loss, vf_loss=vf_loss, entropy=entropy, neglogp=nlp, ratio=ratio, value=v,
                adv_normalised=a), grads


def clip_by_global_norm(grads, max_norm=0.5):
    norm = np.sqrt(sum(float(np.sum(g * g)) for g in grads.values()))
    scale = max_norm / max(norm, max_norm)                   # tf.clip_by_global_norm
    return {k: g * scale for k, g in grads.items()}, norm


class Adam:
    """tf.train.AdamOptimizer(learning_rate=lr, epsilon=1e-5), beta1 0.9, beta2 0.999."""

    def __init__(self, params, lr=3e-4, beta1=0.9, beta2=0.999, epsilon=1e-5):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, epsilon
        self.m = {k: np.zeros_like(v, dtype=np.float64) for k, v in params.items()}
        self.v = {k: np.zeros_like(v, dtype=np.float64) for k, v in params.items()}
        self.t = 0

    def step(self, params, grads):
        self.t += 1
        lr_t = self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for k, g in grads.items():
            self.m[k] = self.b1 * self.m[k] + (1.0 - self.b1) * g
            self.v[k] = self.b2 * self.v[k] + (1.0 - self.b2) * g * g
            params[k] = params[k] - lr_t * self.m[k] / (np.sqrt(self.v[k]) + self.eps)
        return params
