"""Training entry point with the reference's flags (trex_train.py:24-29,113-136), driving the GPU
vec env with the on-device PPO of trex_gym.ppo instead of baselines.ppo2 + TF1.

    python -m trex_gym.trex_train --num_timesteps 5000000 --num_envs 4096

Reward weights are the training ones of the reference (distance 2e2, energy 1e-6, drift 1.0,
trex_train.py:66). --play runs the trained policy and records what a renderer needs per frame (the world poses of the 252
visual meshes); drawing the frames (pybullet's renderer, --debug_render) is out of scope.
"""
import argparse
import os
import sys

import torch

from .ppo import PPO
from .vec_env import TrexVecEnv

_URDF_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "assets", "trex_collide.urdf")


def build_environment(num_envs, device="cuda:0", max_episode_steps=1000):   # (weights of trex_train.py:66)
    return TrexVecEnv(num_envs, urdf_path=_URDF_PATH, device=device, distance_weight=2e2, energy_weight=1e-6,
                      drift_weight=1.0, max_episode_steps=max_episode_steps)


# Hyper-parameter presets. "reference" is the ppo2.learn call of the reference's script (trex_train.py:47-60):
# noptepochs 32, nminibatches 32, lam 0.95, gamma 0.99, lr 3e-4, cliprange 0.2, ent_coef 0. Its rollout is
# nsteps = 4096 samples of ONE env per update; here an update takes nsteps x num_envs samples of the batched env
# (nsteps 32 by default: 131072 samples at 4096 envs, minibatches of 4096). "throughput" is the same except 4 epochs
# per update - what the bench-style measurements of config 3 use.
PRESETS = {
    "reference": dict(nminibatches=32, noptepochs=32, lam=0.95, gamma=0.99, lr=3e-4, cliprange=0.2, ent_coef=0.0),
    "throughput": dict(nminibatches=32, noptepochs=4, lam=0.95, gamma=0.99, lr=3e-4, cliprange=0.2, ent_coef=0.0),
}


def train(env, num_timesteps, seed, nsteps=32, noptepochs=None, save_path=None, log=print, use_graphs=False,
          preset="throughput"):
    hp = dict(PRESETS[preset])
    if noptepochs is not None:
        hp["noptepochs"] = noptepochs
    agent = PPO(env, nsteps=nsteps, seed=seed, use_graphs=use_graphs, **hp)
    log("Number of actions: %d; number of joints: %d; model mass: %.2f; nsteps %d x %d envs; preset %s, noptepochs %d"
        % (env.action_space.shape[0], env.model.num_joints, env.model.total_mass(False), nsteps, env.num_envs, preset,
           hp["noptepochs"]))
    hist = agent.learn(num_timesteps, log=log)
    if save_path:
        st = agent.kern.get_stats()     # VecNormalize's running statistics travel with the weights (trex_train.py:93-99 restores both)
        torch.save({"theta": agent.policy.theta.detach().clone(), "obs_mean": torch.tensor(st["obs_mean"]),
                    "obs_var": torch.tensor(st["obs_var"]), "obs_count": torch.tensor(float(st["obs_count"])),
                    "ret_var": torch.tensor(float(st["ret_var"]))}, save_path)
    return agent, hist


def play(agent, num_play_timesteps, export_path=None, env_index=0, log=print, deterministic=False, update_stats=True, seed=0):
    """The reference's play loop (trex_train.py:126-136: model.step -> env.step -> render a frame -> PNGs -> ffmpeg) up to
    the renderer: every frame's world poses of the 252 visual meshes are recorded (trex_batch_visual_transforms) - what a
    renderer needs to draw the frame; drawing itself is out of scope.
    Defaults = the reference's behaviour: `model.step(obs)[0]` is a SAMPLED action (deterministic=False: N(0, 1) noise for
    the policy kernel) and the VecNormalize env it plays in keeps updating its running statistics with every observation
    (update_stats=True: the statistics kernel runs after each step). deterministic=True plays the mean action,
    update_stats=False freezes the statistics after the reset - the repeatable variant for comparing checkpoints (what this
    function did until round 3, then under the name of the reference's loop).
    export_path: .npz with `mesh_files`, `mesh_links`, `poses` [T, 252, 7] (xyz + quaternion xyzw of env `env_index`),
    `reward` [T], `fps` = 50 (metadata of trex_env.py:36)."""
    import numpy as np
    env, k = agent.env, agent.kern
    noise = torch.zeros(env.num_envs, env.J, device=env.device)
    gen = torch.Generator(device=env.device).manual_seed(int(seed))
    frames, rewards = [], []
    env.reset_tensor()
    k.observe(env.rows, with_reward=False)
    for _ in range(num_play_timesteps):
        if not deterministic:
            noise.normal_(generator=gen)
        k.act(agent.policy.theta, env.rows, noise, agent.actions, clip_obs=agent.clip_obs)
        env.step_tensor(agent.actions)
        if update_stats:
            k.observe(env.rows)       # (VecNormalize.step_wait: the observation moments AND the return moments move on)
        frames.append(env.visual_transforms()[env_index].cpu().numpy())
        rewards.append(float(env.rew[env_index]))
    log("Episode reward: %.3f over %d frames" % (sum(rewards), len(rewards)))
    if export_path:
        table = env.model.visuals()
        np.savez_compressed(export_path, mesh_files=np.array([t[0] for t in table]), mesh_links=np.array([t[1] for t in table]),
                            poses=np.stack(frames).astype(np.float32), reward=np.array(rewards, np.float32), fps=np.float32(50.0))
    return np.stack(frames), np.array(rewards)


def load_agent(load_path, num_envs=1, device="cuda:0", max_episode_steps=1000):
    """trex_train.py:75-110 (replay): rebuild the policy from a file written by train(save_path=...) - the flat
    parameter vector and VecNormalize's statistics."""
    ck = torch.load(load_path, map_location=device, weights_only=True)
    env = build_environment(num_envs, device=device, max_episode_steps=max_episode_steps)
    agent = PPO(env, nsteps=1, nminibatches=1, noptepochs=1)
    with torch.no_grad():
        agent.policy.theta.copy_(ck["theta"])
    st = agent.kern.get_stats()
    st.update(obs_mean=ck["obs_mean"].cpu().numpy(), obs_var=ck["obs_var"].cpu().numpy(), obs_count=float(ck["obs_count"]),
              ret_var=float(ck["ret_var"]))
    agent.kern.set_stats(st)
    return agent


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--train", action="store_true", default=True)
    ap.add_argument("--num_timesteps", type=int, default=int(5e6))   # trex_train.py:27
    ap.add_argument("--random_seed", type=int, default=0)            # trex_train.py:29
    ap.add_argument("--num_envs", type=int, default=4096)
    ap.add_argument("--nsteps", type=int, default=32)
    ap.add_argument("--preset", choices=sorted(PRESETS), default="reference",
                    help="reference: the hyper-parameters of the reference's ppo2.learn call (32 epochs per update)")
    ap.add_argument("--noptepochs", type=int, default=None, help="override the preset's epochs per update")
    ap.add_argument("--max_episode_steps", type=int, default=1000)
    ap.add_argument("--save", type=str, default=None)
    ap.add_argument("--graphs", action="store_true", help="replay the rollout and the minibatch update as HIP graphs")
    ap.add_argument("--play", action="store_true", help="after training: run the policy and record the frames' mesh poses (trex_train.py:25,126-136)")
    ap.add_argument("--num_play_timesteps", type=int, default=int(2e3))          # trex_train.py:28
    ap.add_argument("--export", type=str, default=None, help="with --play: .npz of mesh names + [T, 252, 7] world poses for an external renderer")
    ap.add_argument("--play_deterministic", action="store_true", help="with --play: the mean action and frozen normalisation statistics (the reference samples and keeps updating)")
    args = ap.parse_args(argv)
    env = build_environment(args.num_envs, max_episode_steps=args.max_episode_steps)
    agent, _ = train(env, args.num_timesteps, args.random_seed, args.nsteps, args.noptepochs, args.save, use_graphs=args.graphs,
                     preset=args.preset)
    if args.play:
        play(agent, args.num_play_timesteps, args.export, deterministic=args.play_deterministic, update_stats=not args.play_deterministic)


if __name__ == "__main__":
    sys.exit(main())
