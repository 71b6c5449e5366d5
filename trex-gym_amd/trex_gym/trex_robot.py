"""Robot adapter with the surface of the reference's trex_robot.TrexRobot (trex_robot.py:256-433),
backed by one env of a GPU batch instead of a pybullet body. Only what the env and the training
script read is kept: get_action_limits / get_observation_limits / get_observations /
get_head_position / get_base_position / get_total_joint_power, `_revolute_joint_indices`,
`_total_mass`, `_head_link_index` (trex_train.py:121-123, trex_env.py:93-96,153,187-190)."""
import numpy as np
import torch


class TrexRobot:
    _MAX_JOINT_TORQUE_IN_NM = 300000.0  # trex_robot.py:260

    def __init__(self, vec_env, index=0):
        self._vec = vec_env
        self._index = index
        m = vec_env.model
        # pybullet joint indices of the revolute joints, sorted by joint name (trex_robot.py:311-314)
        self._revolute_joint_indices = list(m.urdf_joint_indices)
        # trex_robot.py:318-320 sums getDynamicsInfo over links 0..n-1, which skips the base link
        self._total_mass = m.total_mass(include_base_link=False)
        self._head_link_index = None  # resolved on the host at model load; kept for attribute parity
        self._starting_configuration = {}

    def _joint_limits(self):
        m = self._vec.model
        return list(m.lower), list(m.upper)

    def get_action_limits(self):
        lo, hi = self._joint_limits()
        return np.array(lo), np.array(hi)

    def get_observation_limits(self):
        lo, hi = self._joint_limits()
        n = len(lo)
        lo.extend([-1.0e12] * 2 * n)
        hi.extend([1.0e12] * 2 * n)
        return np.array(lo), np.array(hi)

    def get_observations(self):
        return self._vec.obs[self._index].cpu().numpy().astype(np.float64).tolist()

    def get_base_position(self):
        return tuple(self._vec.get_state()[self._index, :3].cpu().numpy().astype(np.float64).tolist())

    def get_head_position(self):
        return tuple(self._vec.head_position()[self._index].cpu().numpy().astype(np.float64).tolist())

    def get_total_joint_power(self):
        o = self._vec.obs[self._index]
        J = self._vec.J
        return float(torch.sum(torch.abs(o[J:2 * J] * o[2 * J:])).item())

    def reset(self, reload_urdf=False):
        self._vec.reset_tensor()
