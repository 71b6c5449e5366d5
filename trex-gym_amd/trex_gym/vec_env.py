"""Batched T-rex env on one MI355X: N independent copies of TrexBulletEnv advanced by one HIP kernel
launch per step (C-ABI: include/trex_batch.h).

Surface = baselines' VecEnv, which is what ppo2.learn drives in the reference (trex_train.py:41-49:
DummyVecEnv([make_env]) -> VecNormalize): num_envs, observation_space, action_space, reset(),
step_async(), step_wait(), step(), close(); plus tensor-native variants that keep everything in HBM
(step_tensor) for an on-device policy.  Episodes never terminate in the reference
(should_terminate() is constant False, trex_env.py:183-184); a time limit, if any, is the
harness's: `max_episode_steps` (None = never) auto-resets like a VecEnv does and reports done=True -
inside the step launch itself (trex_batch_set_episode_limit).

Outputs live in ONE row block `rows` [n, 3J+2] f32 = obs | reward | done (written by the kernel in that layout:
trex_batch_step_rows; with penalties_in_rows [n, 3J+5]: the three penalties behind done); `obs`, `rew`, `done_f` are views
into it; `done` holds the flags as bool, `penalties` [n, 3] the three reward terms.

Multi-GPU: one process per GPU, env ids sharded by contiguous range (trex_gym.sharding); the only
exchange is the all-gather of that row block (all_gather_rows / all_gather_rows_pipelined).
"""
import time

import numpy as np
import torch

from . import _capi, sharding, spaces

REWARD_DEFAULTS = dict(distance_weight=1.0, energy_weight=0.005, drift_weight=0.002)  # trex_env.py:42-44


class TrexVecEnv(spaces.Env):       # gym.Env where gym is importable; the surface is baselines' VecEnv
    metadata = {"render.modes": ["human", "rgb_array"], "video.frames_per_second": 50}  # trex_env.py:33-36

    def __init__(self, num_envs, urdf_path=None, collisions_dir=None, device=None, action_repeat=1,
                 distance_weight=1.0, energy_weight=0.005, drift_weight=0.002,
                 max_episode_steps=None, starting_configuration=None, params=None,
                 rank=0, world_size=1, process_group=None, collision="hulls", primitive_max_radius=0.2, row_buffers=1,
                 penalties_in_rows=False):
        """num_envs is the GLOBAL env count; this process owns sharding.shard_range(num_envs, rank, world_size).
        row_buffers=2: successive steps write two row blocks in turn (`rows`, `obs`, `rew`, `done_f` always name the
        block of the LAST step), which lets the pipelined all-gather read a block in place."""
        self.global_num_envs = int(num_envs)
        self.rank, self.world_size, self.process_group = int(rank), int(world_size), process_group
        self.env_lo, self.env_hi = sharding.shard_range(self.global_num_envs, self.rank, self.world_size)
        self.num_envs = self.env_hi - self.env_lo
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _capi.TrexError(-5, "TrexVecEnv needs a HIP device; the physics step has no CPU fallback")
        if self.device.index is None:   # "cuda" without an index means the CURRENT device, for batch and buffers alike
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.model = _capi.Model(urdf_path, collisions_dir)
        # NUM_SUBSTEPS = 5 physics substeps per action repeat (trex_env.py:18,71-73)
        self.model.set_param("substeps", 5 * int(action_repeat))
        for k, v in (params or {}).items():
            self.model.set_param(k, v)
        if collision == "primitives":   # capsules / spheres fitted to the hulls
            self.model.use_primitive_collision(primitive_max_radius)
        elif collision != "hulls":
            raise ValueError("collision must be 'hulls' or 'primitives'")
        for name, angle in (starting_configuration or {}).items():
            self.model.set_start_angle(name, angle)  # unknown name raises, like KeyError at trex_robot.py:307
        self.batch = _capi.Batch(self.model, self.num_envs, self.device.index)
        self.batch.set_reward_weights(distance_weight, energy_weight, drift_weight)
        J = self.J = self.model.num_joints
        lo, hi = self.model.lower.astype(np.float32), self.model.upper.astype(np.float32)
        self.action_space = spaces.Box(low=lo, high=hi, dtype=np.float32)  # trex_robot.py:424-433
        big = np.full(2 * J, 1.0e12, np.float32)                            # trex_robot.py:348-357
        self.observation_space = spaces.Box(low=np.concatenate([lo, -big]), high=np.concatenate([hi, big]),
                                            dtype=np.float32)
        n = self.num_envs
        # obs | reward | done. row_buffers=2: the steps write two blocks in turn, so that a block can be gathered in
        # place while the next step runs (all_gather_rows_pipelined without a staging copy)
        # penalties_in_rows: [n, 3J+5] rows that carry the three penalties behind done (one message for a consumer that
        # wants them); default [n, 3J+2] + a `penalties` tensor of its own. (The 80-column form was MEASURED to cost
        # more HBM write traffic, not less - 8.1 against 3.7 MB per launch of 4096 envs, PMC WRITE_SIZE: DESIGN.md 6.)
        self._pen_in_rows = bool(penalties_in_rows)
        self.batch.set_penalties_in_rows(self._pen_in_rows)      # (explicit: a wide row stride alone writes nothing beyond column 3J + 1)
        self._row_blocks = [torch.zeros(n, 3 * J + (5 if self._pen_in_rows else 2), device=self.device) for _ in range(int(row_buffers))]
        self._penalties = None if self._pen_in_rows else torch.zeros(n, 3, device=self.device)
        self._row_k = 0
        self._point_at(0)
        self.done = torch.zeros(n, dtype=torch.bool, device=self.device)   # the same flags as bytes (written by the kernel too)
        self.max_episode_steps = max_episode_steps
        if max_episode_steps is not None:
            # the step launch itself resets an env whose episode is over (no reset launch between two steps)
            self.batch.set_episode_limit(int(max_episode_steps))
        self._actions = None
        self._gather_buf = None
        self._pipe = None
        self._copy_pipe = None
        self._ep_ret = self._ep_len = None      # episode statistics of the numpy API (step_wait)

    # ---- tensor-native API (stays on device, stream-ordered, no host sync)
    def reset_tensor(self, mask=None):
        """Reset all envs (mask=None) or those with mask != 0 (uint8 or bool [n], e.g. the `done` of the last step).
        Returns obs [n, 3J]; the reward / done columns and `done` of the envs that were reset read 0 afterwards."""
        self.batch.reset_rows(self.rows, mask)
        if mask is None:
            self.done.zero_()
        else:
            self.done.logical_and_(mask == 0)     # (also correct for mask is self.done)
        return self.obs

    @property
    def episode_steps(self):
        """[n] int32: env-steps since each env's last reset (kept by the batch when an episode limit is set)."""
        out = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
        self.batch.get_episode_steps(out)
        return out

    def set_episode_steps(self, steps):
        """Set the per-env step counts (e.g. to stagger the episodes of a benchmark)."""
        self.batch.set_episode_limit(int(self.max_episode_steps or 0), steps.to(device=self.device, dtype=torch.int32).contiguous())

    def step_tensor(self, actions):
        """actions [n, J] f32 on device -> (obs [n,3J], reward [n], done [n] bool). Views into buffers
        that the next call overwrites."""
        if actions.dtype != torch.float32 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        if tuple(actions.shape) != (self.num_envs, self.J):
            raise ValueError("actions must have shape (%d, %d), got %s" % (self.num_envs, self.J, tuple(actions.shape)))
        # (with max_episode_steps the launch also resets the envs whose episode ends with this step: done = 1,
        # reward of the finished step, observation of the new episode - VecEnv semantics, no second launch)
        if len(self._row_blocks) > 1:
            self._point_at(1 - self._row_k)
        self.batch.step_rows(actions, self.rows, self._penalties, done=self.done)   # (None: the penalties ride in the row block)
        return self.obs, self.rew, self.done

    def step_many_tensor(self, actions, rows=None):
        """Open-loop rollout: actions [S, n, J] f32 on device -> rows [S, n, 3J+2] (obs | reward | done of every step), S
        env-steps in ONE launch (trex_batch_step_many; bitwise S calls of step_tensor). `rows`, `obs`, `rew`, `done_f`
        and `done` then hold the last step. sharding.split_rows(rows[s]) cuts a step's block into the three."""
        if actions.dtype != torch.float32 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        if actions.dim() != 3 or tuple(actions.shape[1:]) != (self.num_envs, self.J):
            raise ValueError("actions must have shape (S, %d, %d), got %s" % (self.num_envs, self.J, tuple(actions.shape)))
        S = int(actions.shape[0])
        if rows is None:
            rows = torch.empty(S, self.num_envs, self.rows.shape[1], device=self.device)
        # (the three reward terms of every step: in the rows when penalties_in_rows, else in `penalties_many` [S, n, 3]; `penalties`
        # then holds the last step's, like the other views - it used to keep the values from BEFORE the call)
        self.penalties_many = None if self._pen_in_rows else torch.empty(S, self.num_envs, 3, device=self.device)
        self.batch.step_many(actions, rows, self.penalties_many)
        self.rows.copy_(rows[-1])
        if not self._pen_in_rows:
            self._penalties.copy_(self.penalties_many[-1])
        self.done.copy_(self.done_f != 0)
        return rows

    def _point_at(self, k):
        J = self.J
        self._row_k = k
        self.rows = self._row_blocks[k]
        self.obs, self.rew, self.done_f = self.rows[:, :3 * J], self.rows[:, 3 * J], self.rows[:, 3 * J + 1]
        # lifting_com, station_keeping, energy (trex_env.py:193-195)
        self.penalties = self.rows[:, 3 * J + 2:3 * J + 5] if self._pen_in_rows else self._penalties

    def all_gather_rows(self, rows=None):
        """[global N, 3J+2] = obs | reward | done of EVERY env, on every rank: the one collective of the path
        (RCCL all-gather over xGMI; gloo in the CPU tests). sharding.split_rows(rows, 3 * J) cuts it back."""
        rows = self.rows if rows is None else rows
        if self.world_size == 1:
            return rows
        self._gather_buf = sharding.all_gather_rows(rows, self.global_num_envs, self.world_size,
                                                    self.process_group, out=self._gather_buf)
        return self._gather_buf

    def all_gather_rows_pipelined(self, rows=None, wait=True, join="stream"):
        """Like all_gather_rows, but the collective overlaps the next step: returns the rows gathered by the
        PREVIOUS call (None on the first). See sharding.PipelinedGather."""
        rows = self.rows if rows is None else rows
        if self.world_size == 1:
            return rows
        if self._pipe is None:
            if self.global_num_envs != self.num_envs * self.world_size:
                raise ValueError("pipelined gather needs equal shards")
            self._pipe = sharding.PipelinedGather(self.num_envs, rows.shape[1], self.world_size, rows.dtype,
                                                  self.device, self.process_group)
        return self._pipe.push(rows, copy=not (rows is self.rows and len(self._row_blocks) > 1), wait=wait, join=join)

    def all_gather_rows_copy(self, signal_group=None, sync="barrier"):
        """The pipelined exchange by peer copies instead of a collective kernel (sharding.CopyGather; needs
        row_buffers=2): returns the rows gathered from the PREVIOUS step (None on the first call)."""
        if self.world_size == 1:
            return self.rows
        if len(self._row_blocks) < 2 or self.global_num_envs != self.num_envs * self.world_size:
            raise ValueError("the copy exchange reads the row block in place: row_buffers=2 and equal shards")
        if self._copy_pipe is None:
            self._copy_pipe = sharding.CopyGather(self.num_envs, self.rows.shape[1], self.world_size, self.rank, self.rows.dtype,
                                                  self.device, self.process_group, signal_group, sync)
        return self._copy_pipe.push(self.rows)

    def all_gather_obs(self):
        """[global N, 3J]: the observation columns of all_gather_rows()."""
        return self.all_gather_rows()[:, :3 * self.J]

    # ---- baselines VecEnv API (host numpy in/out)
    def reset(self):
        if self._ep_ret is not None:
            self._ep_ret[:] = 0.0
            self._ep_len[:] = 0
        return self.reset_tensor().cpu().numpy()

    def step_async(self, actions):
        self._actions = torch.as_tensor(np.asarray(actions, np.float32)).to(self.device, non_blocking=True)

    def step_wait(self):
        """-> obs, rewards, dones, infos as numpy / dicts. The reference wraps its env in baselines' bench.Monitor
        (trex_train.py:41-42), whose info['episode'] = {'r': episode return, 'l': length, 't': seconds since the monitor was
        made} is what ppo2.learn averages into eprewmean / eplenmean: the env whose episode ENDS with this step (the harness's
        time limit inside the launch, or a contained env) carries that entry here too."""
        obs, rew, done = self.step_tensor(self._actions)
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        if self._ep_ret is None:
            self._ep_ret, self._ep_len = np.zeros(self.num_envs, np.float64), np.zeros(self.num_envs, np.int64)
            self._t_start = time.time()
        self._ep_ret += rew          # (Monitor sums Python floats: f64)
        self._ep_len += 1
        infos = [{} for _ in range(self.num_envs)]
        for i in np.flatnonzero(done):
            infos[i]["episode"] = {"r": round(float(self._ep_ret[i]), 6), "l": int(self._ep_len[i]),
                                   "t": round(time.time() - self._t_start, 6)}
            self._ep_ret[i] = 0.0
            self._ep_len[i] = 0
        return obs, rew, done, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        # the batch caches the device allocations it has validated; the buffers of this env go back to torch's allocator now
        self.batch.forget_buffers()
        self.batch.close()

    def render(self, mode="rgb_array"):
        return np.array([])  # rendering is outside the accelerated path

    def seed(self, seed=None):
        return [seed]  # the env is deterministic; np_random is never consumed (trex_env.py:124-126)

    # ---- state access for tests / checkpointing
    def get_state(self):
        out = torch.zeros(self.num_envs, self.batch.state_width, device=self.device)
        self.batch.get_state(out)
        return out

    def set_state(self, state, motors_enabled=True):
        state = state.to(device=self.device, dtype=torch.float32).contiguous()
        assert tuple(state.shape) == (self.num_envs, self.batch.state_width)
        self.batch.set_state(state)
        self.batch.set_motors_enabled(motors_enabled)

    def head_position(self):
        out = torch.zeros(self.num_envs, 3, device=self.device)
        self.batch.head_position(out)
        return out

    def link_transforms(self):
        """World pose of every URDF link frame, [n, L, 7] = xyz + quaternion xyzw (rollout export for
        rendering, the step after the path: trex_env.py:156-181)."""
        L = len(self.model.links())
        out = torch.empty(self.num_envs, L, 7, device=self.device)
        self.batch.link_transforms(out)
        return out

    def visual_transforms(self):
        """World pose of every <visual> mesh of the URDF, [n, V, 7] = xyz + quaternion xyzw (252 meshes for trex.urdf;
        `model.visuals()` names them): what a renderer places the meshes with - the table pybullet keeps for
        getCameraImage (trex_env.py:164-176)."""
        V = len(self.model.visuals())
        out = torch.empty(self.num_envs, V, 7, device=self.device)
        self.batch.visual_transforms(out)
        return out

    def set_domain(self, mass_scale=None, friction=None):
        """Per-env domain randomisation (BASELINE config 5): mass_scale [n, num_bodies], friction [n]."""
        if mass_scale is not None:
            mass_scale = mass_scale.to(device=self.device, dtype=torch.float32).contiguous()
            assert tuple(mass_scale.shape) == (self.num_envs, self.model.num_bodies)
        if friction is not None:
            friction = friction.to(device=self.device, dtype=torch.float32).contiguous()
            assert tuple(friction.shape) == (self.num_envs,)
        self.batch.set_domain(mass_scale, friction)
