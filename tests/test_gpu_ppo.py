"""BASELINE config 3: PPO driving the vectorised env entirely on device (trex_gym.ppo)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_ppo_runs_and_improves_the_return():
    from trex_gym import trex_train
    env = trex_train.build_environment(1024, max_episode_steps=200)
    logs = []
    agent, hist = trex_train.train(env, num_timesteps=1024 * 32 * 12, seed=0, nsteps=32, noptepochs=4, log=logs.append)
    assert len(hist) == 12
    for h in hist:
        assert all(math.isfinite(v) for v in (h["policy_loss"], h["value_loss"], h["entropy"], h["mean_step_reward"]))
    # the policy learns to stop flailing: the mean per-step reward of the last iterations beats the first
    first = sum(h["mean_step_reward"] for h in hist[:3]) / 3
    last = sum(h["mean_step_reward"] for h in hist[-3:]) / 3
    assert last > first, (first, last)
    assert hist[-1]["env_steps_per_s"] > 1e5       # north-star floor, with the learner in the loop
    assert torch.isfinite(agent.obs).all()


def test_running_mean_std_matches_batch_statistics():
    from trex_gym.ppo import RunningMeanStd
    g = torch.Generator(device="cuda:0").manual_seed(0)
    x = 3 + 2 * torch.randn(5000, 7, device="cuda:0", generator=g)
    r = RunningMeanStd((7,), "cuda:0")
    for chunk in x.split(617):
        r.update(chunk)
    assert torch.allclose(r.mean.float(), x.mean(0), atol=1e-3)
    assert torch.allclose(r.var.float(), x.var(0, unbiased=False), rtol=1e-3)
