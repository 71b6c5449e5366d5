/*
 * trex_policy.h - C-ABI of the trainer-side per-step work that sits on either side of the batched physics step
 * when a PPO2-style trainer drives it (the reference does this in Python/TF through baselines:
 * VecNormalize + MlpPolicy.step + the Runner's GAE, trex_train.py:41-61).
 *
 * Not part of the physics boundary (include/trex_batch.h): a caller that brings its own policy never needs it.
 * It exists because at 11 M env-steps/s the ~30 small framework kernels per step of a stock policy halve the
 * rollout rate: the whole per-step caller work is two launches here.
 *
 *   trex_policy_observe   VecNormalize.step_wait (trex_train.py:45): ret = ret*gamma + rew; running mean / variance of
 *                         the observations and of the returns (Chan's parallel update, f64 state); ret[done] = 0.
 *   trex_policy_act       MlpPolicy.step (trex_train.py:48): normalise + clip the observation, both 2 x 64 tanh MLPs
 *                         (policy mean, value) on the matrix cores (v_mfma_f32_32x32x2_f32: exact f32), Gaussian
 *                         sample from caller-supplied N(0,1) noise, log-probability; writes the env's action rows and
 *                         the rollout buffers.
 *   trex_policy_gae       the Runner's GAE(lambda) over a [T, N] rollout (reward normalisation applied on the way).
 *   trex_policy_adam      global-norm gradient clip + Adam in TensorFlow's form (what ppo2's trainer applies).
 *
 * Conventions as in trex_batch.h: 0 / negative TREX_E_* codes, trex_last_error(), caller-owned device buffers that
 * are validated once per allocation, stream-ordered asynchronous launches (void* = hipStream_t), graph-capturable.
 *
 * Flat parameter vector `theta` (f32, trex_policy_param_count() elements), weights stored [in][out]:
 *   pi.W1 [D,H] pi.b1 [H] pi.W2 [H,H] pi.b2 [H] pi.W3 [H,A] pi.b3 [A]
 *   vf.W1 [D,H] vf.b1 [H] vf.W2 [H,H] vf.b2 [H] vf.W3 [H,1] vf.b3 [1]   logstd [A]
 * with D = obs_dim, A = act_dim, H = hidden = 64; every block starts on a multiple of 4 floats (a few unused pad
 * elements follow the blocks whose size is not one); trex_policy_param_offsets() returns the 13 offsets in this order.
 */
#ifndef TREX_POLICY_H
#define TREX_POLICY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

typedef struct TrexPolicy TrexPolicy;

/* hidden must be 64, act_dim <= 32, obs_dim <= 126. Owns the normalisation state of num_envs envs. */
int trex_policy_create(int num_envs, int obs_dim, int act_dim, int hidden, int device, TrexPolicy **out);
void trex_policy_destroy(TrexPolicy *policy);
int trex_policy_param_count(const TrexPolicy *policy);
int trex_policy_param_offsets(const TrexPolicy *policy, int offsets[13]);

/* VecNormalize state. stats (host, f64): obs mean [D], obs var [D], obs count, ret mean, ret var, ret count,
 * sum of raw rewards seen since the last set (for logging) = 2D + 5 doubles. Synchronises the stream. */
int trex_policy_get_stats(TrexPolicy *policy, double *stats_host, void *stream);
int trex_policy_set_stats(TrexPolicy *policy, const double *stats_host, void *stream);
/* discounted return per env, [N] f32 device (VecNormalize.ret) */
int trex_policy_get_returns(TrexPolicy *policy, float *ret_dev, void *stream);

/* After an env step (or a reset: with_reward = 0) wrote the row block rows_dev [N, row_stride] = obs | reward | done
 * (trex_batch_step_rows): update the running statistics with this batch of observations and, with_reward != 0, of
 * returns. Optional outputs (device, nullable): raw_rew_out [N] and done_out [N] (this step's columns, for the
 * rollout buffers), rew_scale_out [1] = 1 / sqrt(ret var + 1e-8) AFTER this update (what VecNormalize divides this
 * step's rewards by). gamma: the return discount (VecNormalize's, 0.99). */
int trex_policy_observe(TrexPolicy *policy, const float *rows_dev, int row_stride, int with_reward, float gamma,
                        float *raw_rew_out, float *done_out, float *rew_scale_out, void *stream);

/* One policy evaluation for all N envs on the observations in rows_dev (raw; normalised here with the current
 * statistics and clipped to +-clip_obs):
 *   noise_dev    [N, A] f32, N(0,1) draws (the caller's generator; nullable only with value_only)
 *   actions_dev  [N, A] out: mean + exp(logstd) * noise - what trex_batch_step[_rows] takes (it clips to the limits)
 *   obs_out [N, D], act_out [N, A], logp_out [N], value_out [N]: rollout-buffer rows of this step, each nullable
 *   value_only != 0: only value_out is written (the bootstrap value after the last step). */
int trex_policy_act(TrexPolicy *policy, const float *theta_dev, const float *rows_dev, int row_stride, float clip_obs,
                    const float *noise_dev, float *actions_dev, float *obs_out, float *act_out, float *logp_out,
                    float *value_out, int value_only, void *stream);

/* GAE(lambda) over a rollout: raw_rew [T, N], rew_scale [T], done [T, N] (the flag RETURNED by step t), values
 * [T + 1, N] (row T = value of the observation after the last step) -> adv [T, N], ret [T, N] = adv + values;
 * rewards enter as clip(raw * scale[t], +-clip_rew). */
int trex_policy_gae(TrexPolicy *policy, const float *raw_rew_dev, const float *rew_scale_dev, const float *done_dev,
                    const float *values_dev, float *adv_dev, float *ret_dev, int T, float gamma, float lam,
                    float clip_rew, void *stream);

/* One optimiser step on the flat vectors (all [P] f32 device; m, v: Adam moments, zero-initialised by the caller):
 * g = grad * max_norm / max(|grad|, max_norm); m = b1 m + (1 - b1) g; v = b2 v + (1 - b2) g^2;
 * theta -= lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps)   (tf.train.AdamOptimizer's form);
 * then grad is zeroed for the next accumulation. The step count t lives in the policy object (reset: _adam_reset).
 * grad_norm_out [1] f32 device, nullable. */
int trex_policy_adam(TrexPolicy *policy, float *theta_dev, float *grad_dev, float *m_dev, float *v_dev, float lr,
                     float beta1, float beta2, float eps, float max_grad_norm, float *grad_norm_out, void *stream);
int trex_policy_adam_reset(TrexPolicy *policy, void *stream);

/* ---- the learner: one PPO2 minibatch step (ppo2's Model.train, trex_train.py:49-61) as three launches ----
 * The rollout buffers are flat [num_samples, ...] f32 device arrays (num_samples = T x N): obs (NORMALISED, as the
 * policy saw them), act, logp (log-probability under the rollout policy), val (its value estimates), adv, ret.
 * perm_dev [>= first + mb] i64 device: a permutation of the sample indices (an epoch's shuffle); the minibatch is
 * perm[first .. first + mb).
 *
 * trex_policy_minibatch_stats: for every minibatch k of an epoch (perm[k mb .. (k + 1) mb)) the mean of its
 *   advantages and 1 / (std + 1e-8) (population std, as numpy's) -> stats_out [num_minibatches, 2] f32 device.
 * trex_policy_minibatch_step: forward of both MLPs on the matrix cores, clipped surrogate (cliprange) + clipped value
 *   loss (vf_coef) - ent_coef * entropy, analytic backward pass, global-norm clip (max_grad_norm), Adam in
 *   TensorFlow's form (lr, beta1, beta2, eps) on theta / m / v [P]; the step count is the policy object's
 *   (trex_policy_adam_reset). grad_dev [P]: receives the UNclipped gradient of this minibatch (scratch; tests read it).
 *   adv_stats_dev [2]: this minibatch's row of the stats above. loss_sums_dev [2] f32 device, nullable: the
 *   minibatch's mean surrogate loss and mean value loss are ADDED to it (zero it per update; divide by the steps).
 *   obs_dim <= 96 here (TREX_E_INVALID beyond: the kernel stages 96 observation columns per sample). */
int trex_policy_minibatch_stats(TrexPolicy *policy, const float *adv_dev, int64_t num_samples, const int64_t *perm_dev,
                                int num_minibatches, int mb, float *stats_out_dev, void *stream);
/* trex_policy_minibatch_grad: the first two launches of trex_policy_minibatch_step alone - the minibatch's gradient
 *   (entropy term included, UNclipped) times grad_scale -> grad_dev [P]; no clip, no Adam, the step count does not move.
 *   For a DATA-PARALLEL trainer (one process per GPU, each with its own envs and rollout; SURVEY 8e: "only PPO gradient
 *   all-reduce remains"): every rank calls this with grad_scale = 1 / ranks on its share of the minibatch, the ranks SUM
 *   grad_dev (torch.distributed.all_reduce over RCCL), then every rank applies trex_policy_adam - the update of the
 *   concatenated minibatch, on every rank the same. adv_stats must then be the statistics of the WHOLE minibatch (all
 *   ranks' shares; trex_gym/ppo.py forms them). loss_sums_dev as above, scaled by grad_scale too. */
int trex_policy_minibatch_grad(TrexPolicy *policy, const float *theta_dev, float *grad_dev, const float *obs_dev,
                               const float *act_dev, const float *logp_dev, const float *val_dev, const float *adv_dev,
                               const float *ret_dev, int64_t num_samples, const int64_t *perm_dev, int first, int mb,
                               const float *adv_stats_dev, float cliprange, float ent_coef, float vf_coef, float grad_scale,
                               float *loss_sums_dev, void *stream);
int trex_policy_minibatch_step(TrexPolicy *policy, float *theta_dev, float *grad_dev, float *m_dev, float *v_dev,
                               const float *obs_dev, const float *act_dev, const float *logp_dev, const float *val_dev,
                               const float *adv_dev, const float *ret_dev, int64_t num_samples, const int64_t *perm_dev,
                               int first, int mb, const float *adv_stats_dev, float cliprange, float ent_coef, float vf_coef,
                               float lr, float beta1, float beta2, float eps, float max_grad_norm, float *loss_sums_dev,
                               void *stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* TREX_POLICY_H */
