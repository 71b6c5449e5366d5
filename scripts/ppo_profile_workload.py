import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "trex-gym_amd"))
from trex_gym.ppo import PPO
from trex_gym.trex_train import build_environment
env = build_environment(4096)
agent = PPO(env, nsteps=32, nminibatches=32, noptepochs=1, seed=0, use_graphs=False)
for _ in range(4):
    b = agent.collect()
agent.update(b)
torch.cuda.synchronize()
