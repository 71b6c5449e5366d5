"""TrexBulletEnv with the reference's gym surface (trex_env.py:25-196), one env on the GPU batch.

Same constructor keywords, spaces, old-gym 4-tuple step(), python-list observations. The physics
engine behind it is the HIP kernel (C-ABI include/trex_batch.h), not pybullet. For throughput use
trex_gym.vec_env.TrexVecEnv; this single-env class exists so code written against the reference
(`trex_gym.trex_env.TrexBulletEnv(urdf_path)`) runs unmodified.
"""
import numpy as np
import torch

from . import spaces, trex_robot
from .vec_env import TrexVecEnv

NUM_SUBSTEPS = 5
FLOOR_URDF_FILENAME = 'floor.urdf'
EARTH_GRAVITATIONAL_CONSTANT = 9.81
RENDER_HEIGHT = 720
RENDER_WIDTH = 960


class TrexBulletEnv(spaces.Env):     # gym.Env where gym is importable (trex_env.py:25)
    metadata = {
        "render.modes": ["human", "rgb_array"],
        "video.frames_per_second": 50
    }

    def __init__(self, urdf_path=None, action_repeat=1, distance_weight=1.0, energy_weight=0.005,
                 drift_weight=0.002, render=False, device=None):
        if render:
            raise NotImplementedError("rendering is outside the accelerated path (DESIGN.md, out of scope)")
        self._time_step = 0.01 / NUM_SUBSTEPS
        self._urdf_path = urdf_path
        self._action_repeat = action_repeat * NUM_SUBSTEPS
        self._num_bullet_solver_iterations = 300 // NUM_SUBSTEPS
        self._observation = []
        self._env_step_counter = 0
        self._is_render = render
        self._last_base_position = [0.] * 3
        self._weight_distance = distance_weight
        self._weight_energy = energy_weight
        self._weight_drift = drift_weight
        self._action_bound = 1
        # the v1 env spells the joints 'femur_L_joint' (trex_env.py:81-87); the loader accepts both
        self._starting_configuration = {'femur_L_joint': -0.6, 'tibia_L_joint': 0.4,
                                        'tarsometatarsus_L_joint': -1.2, 'femur_R_joint': -0.6,
                                        'tibia_R_joint': 0.4, 'tarsometatarsus_R_joint': -1.2}
        self._vec = TrexVecEnv(1, urdf_path=urdf_path, device=device, action_repeat=action_repeat,
                               distance_weight=distance_weight, energy_weight=energy_weight,
                               drift_weight=drift_weight,
                               starting_configuration=self._starting_configuration)
        self.model = trex_robot.TrexRobot(self._vec, 0)
        self.np_random = None
        self.seed()
        self.reset()
        action_low, action_high = self.model.get_action_limits()
        self.action_space = spaces.Box(low=action_low, high=action_high, dtype=np.float32)
        observation_low, observation_high = self.model.get_observation_limits()
        self.observation_space = spaces.Box(low=observation_low, high=observation_high, dtype=np.float32)

    def reset(self):
        self._vec.reset_tensor()
        self._env_step_counter = 0
        self._last_base_position = self.model.get_base_position()
        return self.model.get_observations()

    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)  # created, never consumed (trex_env.py:124-126)
        return [seed]

    def step(self, action):
        action = np.asarray(action, dtype=np.float32).reshape(-1)
        if action.shape[0] < self._vec.J:
            raise ValueError("The action dimension is not the same as the number of motors.")
        # only the first J entries are joint targets (trex_robot.py:418-421); the kernel clips
        a = torch.from_numpy(action[: self._vec.J].copy()).reshape(1, -1)
        self._vec.step_tensor(a)
        self._env_step_counter += 1
        self._observation = self.model.get_observations()
        return self._observation, self.compute_reward(), self.should_terminate(), {}

    def render(self, mode='rgb_array', close=False):
        return np.array([])

    def should_terminate(self):
        return False

    def compute_reward(self):
        # computed inside the step kernel (trex_env.py:186-196); penalties = the three logged values
        self._last_penalties = dict(zip(("penalty_lifting_com", "penalty_station_keeping", "penalty_energy"),
                                        self._vec.penalties[0].cpu().numpy().astype(np.float64).tolist()))
        return float(self._vec.rew[0].item())

    def close(self):
        self._vec.close()
