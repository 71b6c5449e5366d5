"""ctypes binding of libtrex_hip.so (C-ABI in include/trex_batch.h).

There is NO CPU fallback: if the HIP library is missing, importing this module raises; if no GPU
is present, creating a batch raises TrexError (model loading is host-only and still works).
"""
import ctypes as C
import os

import numpy as np
import torch  # noqa: F401  - FIRST: torch brings its own HIP runtime; loading libtrex_hip.so before it
#                             would bind the system runtime and leave the process with two of them

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TREX_LIB") or os.path.join(_HERE, "libtrex_hip.so")  # TREX_LIB: diagnostic builds

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "trex_gym: %s not found - build it with `python __graft_entry__.py` or "
        "`make -C trex-gym_amd/csrc` (hipcc --offload-arch=gfx950). The physics step has no CPU fallback."
        % LIB_PATH)

lib = C.CDLL(LIB_PATH)

PARAM_NAMES = ["dt", "substeps", "iterations", "gravity", "motor_kp", "motor_kd", "motor_max_force",
               "floor_z", "friction", "erp", "contact_erp", "contact_margin", "link_damping",
               "max_coordinate_velocity", "max_contacts"]
# every symbol include/trex_batch.h declares (tests/test_capi_symbols.py parses the header too)
SYMBOLS = [
    "trex_last_error", "trex_model_load", "trex_model_destroy", "trex_model_num_bodies",
    "trex_model_num_joints", "trex_model_num_urdf_joints", "trex_model_num_hull_vertices",
    "trex_model_total_mass", "trex_model_joint_info", "trex_model_set_start_angle",
    "trex_model_set_start_pose", "trex_model_set_param", "trex_model_get_param", "trex_model_get_array",
    "trex_batch_create", "trex_batch_destroy", "trex_batch_num_envs", "trex_batch_set_reward_weights",
    "trex_batch_reset", "trex_batch_step", "trex_batch_get_state", "trex_batch_set_state",
    "trex_batch_set_motors_enabled", "trex_batch_head_position", "trex_batch_set_domain",
    "trex_batch_contact_stats", "trex_batch_debug_step", "trex_batch_launch_info", "trex_batch_time_steps",
    "trex_model_num_links", "trex_model_link_info", "trex_batch_link_transforms",
    "trex_model_use_primitive_collision", "trex_model_fit_hull_primitives",
    "trex_build_id", "trex_batch_step_rows", "trex_batch_reset_rows",
    "trex_batch_set_episode_limit", "trex_batch_get_episode_steps",
    "trex_batch_set_wave_balance", "trex_batch_forget_buffers", "trex_batch_set_penalties_in_rows",
    "trex_model_num_visuals", "trex_model_visual_info", "trex_batch_visual_transforms", "trex_batch_step_many",
]

# every symbol include/trex_policy.h declares (the trainer-side kernels)
POLICY_SYMBOLS = [
    "trex_policy_create", "trex_policy_destroy", "trex_policy_param_count", "trex_policy_param_offsets",
    "trex_policy_get_stats", "trex_policy_set_stats", "trex_policy_get_returns", "trex_policy_observe",
    "trex_policy_act", "trex_policy_gae", "trex_policy_adam", "trex_policy_adam_reset",
    "trex_policy_minibatch_stats", "trex_policy_minibatch_step", "trex_policy_minibatch_grad",
]

_vp = C.c_void_p
lib.trex_last_error.restype = C.c_char_p
lib.trex_build_id.restype = C.c_char_p
lib.trex_model_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(_vp)]
lib.trex_model_destroy.argtypes = [_vp]
lib.trex_model_destroy.restype = None
for _n in ("trex_model_num_bodies", "trex_model_num_joints", "trex_model_num_urdf_joints",
           "trex_model_num_hull_vertices"):
    getattr(lib, _n).argtypes = [_vp]
lib.trex_model_total_mass.argtypes = [_vp, C.c_int]
lib.trex_model_total_mass.restype = C.c_double
lib.trex_model_joint_info.argtypes = [_vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int),
                                      C.POINTER(C.c_double), C.POINTER(C.c_double)]
lib.trex_model_set_start_angle.argtypes = [_vp, C.c_char_p, C.c_double]
lib.trex_model_set_start_pose.argtypes = [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
lib.trex_model_set_param.argtypes = [_vp, C.c_char_p, C.c_double]
lib.trex_model_get_param.argtypes = [_vp, C.c_char_p, C.POINTER(C.c_double)]
lib.trex_model_get_array.argtypes = [_vp, C.c_char_p, C.POINTER(C.c_double), C.c_int]
lib.trex_batch_create.argtypes = [_vp, C.c_int, C.c_int, C.POINTER(_vp)]
lib.trex_batch_destroy.argtypes = [_vp]
lib.trex_batch_destroy.restype = None
lib.trex_batch_num_envs.argtypes = [_vp]
lib.trex_batch_set_reward_weights.argtypes = [_vp, C.c_float, C.c_float, C.c_float]
lib.trex_batch_reset.argtypes = [_vp, _vp, _vp, _vp]
lib.trex_batch_step.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp]
lib.trex_batch_step_rows.argtypes = [_vp, _vp, _vp, C.c_int, _vp, _vp, _vp]
lib.trex_batch_step_many.argtypes = [_vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]
lib.trex_batch_reset_rows.argtypes = [_vp, _vp, _vp, C.c_int, _vp]
lib.trex_batch_set_episode_limit.argtypes = [_vp, C.c_int, _vp, _vp]
lib.trex_batch_get_episode_steps.argtypes = [_vp, _vp, _vp]
lib.trex_batch_set_wave_balance.argtypes = [_vp, C.c_int]
lib.trex_batch_forget_buffers.argtypes = [_vp]
lib.trex_batch_set_penalties_in_rows.argtypes = [_vp, C.c_int]
lib.trex_batch_debug_step.argtypes = [_vp, _vp, _vp, _vp, _vp]
lib.trex_batch_get_state.argtypes = [_vp, _vp, _vp]
lib.trex_batch_set_state.argtypes = [_vp, _vp, _vp]
lib.trex_batch_set_motors_enabled.argtypes = [_vp, C.c_int, _vp]
lib.trex_batch_head_position.argtypes = [_vp, _vp, _vp]
lib.trex_batch_set_domain.argtypes = [_vp, _vp, _vp, _vp]
lib.trex_model_use_primitive_collision.argtypes = [_vp, C.c_double, C.c_int, C.c_int]
lib.trex_model_fit_hull_primitives.argtypes = [_vp, C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int]
lib.trex_model_num_links.argtypes = [_vp]
lib.trex_model_link_info.argtypes = [_vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int)]
lib.trex_batch_link_transforms.argtypes = [_vp, _vp, _vp]
lib.trex_model_num_visuals.argtypes = [_vp]
lib.trex_model_visual_info.argtypes = [_vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
lib.trex_batch_visual_transforms.argtypes = [_vp, _vp, _vp]
lib.trex_batch_contact_stats.argtypes = [_vp, _vp, _vp, _vp]
lib.trex_batch_launch_info.argtypes = [_vp] + [C.POINTER(C.c_int)] * 4
lib.trex_batch_time_steps.argtypes = [_vp, _vp, _vp, _vp, _vp, C.c_int, _vp, C.POINTER(C.c_float)]


lib.trex_policy_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]
lib.trex_policy_destroy.argtypes = [_vp]
lib.trex_policy_destroy.restype = None
lib.trex_policy_param_count.argtypes = [_vp]
lib.trex_policy_param_offsets.argtypes = [_vp, C.POINTER(C.c_int)]
lib.trex_policy_get_stats.argtypes = [_vp, C.POINTER(C.c_double), _vp]
lib.trex_policy_set_stats.argtypes = [_vp, C.POINTER(C.c_double), _vp]
lib.trex_policy_get_returns.argtypes = [_vp, _vp, _vp]
lib.trex_policy_observe.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_float, _vp, _vp, _vp, _vp]
lib.trex_policy_act.argtypes = [_vp, _vp, _vp, C.c_int, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp]
lib.trex_policy_gae.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_float, C.c_float, C.c_float, _vp]
lib.trex_policy_adam.argtypes = [_vp, _vp, _vp, _vp, _vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _vp, _vp]
lib.trex_policy_adam_reset.argtypes = [_vp, _vp]
lib.trex_policy_minibatch_stats.argtypes = [_vp, _vp, C.c_int64, _vp, C.c_int, C.c_int, _vp, _vp]
lib.trex_policy_minibatch_step.argtypes = ([_vp] * 11 + [C.c_int64, _vp, C.c_int, C.c_int, _vp] + [C.c_float] * 8 + [_vp, _vp])
lib.trex_policy_minibatch_grad.argtypes = ([_vp] * 9 + [C.c_int64, _vp, C.c_int, C.c_int, _vp] + [C.c_float] * 4 + [_vp, _vp])


class TrexError(RuntimeError):
    """Raised where the reference would see pybullet.error / KeyError from the engine boundary."""

    def __init__(self, code, message):
        super().__init__("%s (code %d)" % (message, code))
        self.code = code


def check(code):
    if code < 0:
        raise TrexError(code, lib.trex_last_error().decode())
    return code


E_INVALID = -1   # TREX_E_INVALID


def build_id():
    return lib.trex_build_id().decode()


def _ptr(t, device=None, dtype=None, numel=None, what="tensor"):
    """torch tensor / None -> void* (device pointer). A host tensor, a tensor on another GPU, of another
    dtype or too short would be misread or fault the GPU, so it is refused here (TREX_E_INVALID); the
    C-ABI checks the raw pointer again (capi.cpp check_device_buffer)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise TrexError(E_INVALID, "%s: expected a tensor in device memory, got one on %s" % (what, t.device))
    if not t.is_contiguous():
        raise TrexError(E_INVALID, "%s: expected a contiguous tensor" % what)
    if device is not None and t.device.index != device:
        raise TrexError(E_INVALID, "%s: tensor on cuda:%s, batch on cuda:%d" % (what, t.device.index, device))
    if dtype is not None and t.dtype != dtype:
        raise TrexError(E_INVALID, "%s: expected dtype %s, got %s" % (what, dtype, t.dtype))
    if numel is not None and t.numel() < numel:
        raise TrexError(E_INVALID, "%s: expected at least %d elements, got %d" % (what, numel, t.numel()))
    return C.c_void_p(t.data_ptr())


def default_urdf_path():
    return os.path.join(_HERE, "..", "assets", "trex_collide.urdf")


class Model:
    """Compiled model (host side). Replaces loadURDF + joint/dynamics introspection."""

    def __init__(self, urdf_path=None, collisions_dir=None):
        self.urdf_path = os.path.abspath(urdf_path or default_urdf_path())
        h = _vp()
        check(lib.trex_model_load(self.urdf_path.encode(),
                                  collisions_dir.encode() if collisions_dir else None, C.byref(h)))
        self.h = h
        self.num_bodies = lib.trex_model_num_bodies(h)
        self.num_joints = lib.trex_model_num_joints(h)
        self.num_urdf_joints = lib.trex_model_num_urdf_joints(h)
        self.joint_names, self.urdf_joint_indices, lo, hi = [], [], [], []
        for k in range(self.num_joints):
            name, idx, l, u = C.c_char_p(), C.c_int(), C.c_double(), C.c_double()
            check(lib.trex_model_joint_info(h, k, C.byref(name), C.byref(idx), C.byref(l), C.byref(u)))
            self.joint_names.append(name.value.decode())
            self.urdf_joint_indices.append(idx.value)
            lo.append(l.value)
            hi.append(u.value)
        self.lower = np.array(lo)
        self.upper = np.array(hi)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib.trex_model_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def links(self):
        """[(name, body index)] of every URDF link, document order."""
        out = []
        for k in range(lib.trex_model_num_links(self.h)):
            name, body = C.c_char_p(), C.c_int()
            check(lib.trex_model_link_info(self.h, k, C.byref(name), C.byref(body)))
            out.append((name.value.decode(), body.value))
        return out

    def visuals(self):
        """[(mesh file, link index, xyz[3], quat_xyzw[4])] of every <visual> mesh, document order
        (tools/urdf_parsing.py:93-120: UrdfLink.visual_shapes)."""
        out = []
        for k in range(lib.trex_model_num_visuals(self.h)):
            name, link = C.c_char_p(), C.c_int()
            xyz, quat = (C.c_double * 3)(), (C.c_double * 4)()
            check(lib.trex_model_visual_info(self.h, k, C.byref(name), C.byref(link), xyz, quat))
            out.append((name.value.decode(), link.value, np.array(xyz[:]), np.array(quat[:])))
        return out

    def use_primitive_collision(self, max_radius=0.2, max_divisions=3, min_points=4):
        """Replace the convex hulls by fitted capsules / spheres (tools/mesh_primitives.py:323-402)."""
        check(lib.trex_model_use_primitive_collision(self.h, float(max_radius), int(max_divisions), int(min_points)))

    def fit_hull_primitives(self, group, max_radius=0.2, max_divisions=3, min_points=4):
        """[(p0, p1, radius)] fitted to convex hull `group` (body frame); p0 == p1 for a sphere."""
        n = check(lib.trex_model_fit_hull_primitives(self.h, group, max_radius, max_divisions, min_points, None, 0))
        out = np.zeros((n, 7))
        check(lib.trex_model_fit_hull_primitives(self.h, group, max_radius, max_divisions, min_points,
                                                 out.ctypes.data_as(C.POINTER(C.c_double)), n))
        return [(o[0:3].copy(), o[3:6].copy(), float(o[6])) for o in out]

    def total_mass(self, include_base_link=False):
        return lib.trex_model_total_mass(self.h, int(include_base_link))

    def set_start_angle(self, joint_name, angle):
        check(lib.trex_model_set_start_angle(self.h, joint_name.encode(), float(angle)))

    def set_start_pose(self, xyz, rpy):
        a = (C.c_double * 3)(*xyz)
        b = (C.c_double * 3)(*rpy)
        check(lib.trex_model_set_start_pose(self.h, a, b))

    def set_param(self, name, value):
        check(lib.trex_model_set_param(self.h, name.encode(), float(value)))

    def get_param(self, name):
        v = C.c_double()
        check(lib.trex_model_get_param(self.h, name.encode(), C.byref(v)))
        return v.value

    def array(self, name):
        n = check(lib.trex_model_get_array(self.h, name.encode(), None, 0))
        out = np.zeros(n)
        check(lib.trex_model_get_array(self.h, name.encode(), out.ctypes.data_as(C.POINTER(C.c_double)), n))
        return out


class Batch:
    """N env copies on one GPU; all tensors are torch CUDA(HIP) tensors owned by the caller."""

    def __init__(self, model, num_envs, device=0):
        self.model = model
        self.num_envs = int(num_envs)
        self.device = int(device)
        h = _vp()
        check(lib.trex_batch_create(model.h, self.num_envs, self.device, C.byref(h)))
        self.h = h
        self.J = model.num_joints
        self.state_width = 13 + 2 * self.J

    def close(self):
        if getattr(self, "h", None):
            lib.trex_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _stream(stream):
        if stream is None:
            import torch
            return C.c_void_p(torch.cuda.current_stream().cuda_stream)
        return C.c_void_p(stream)

    def set_reward_weights(self, distance, energy, drift):
        check(lib.trex_batch_set_reward_weights(self.h, distance, energy, drift))

    def _p(self, t, dtype, numel, what):
        import torch
        return _ptr(t, self.device, getattr(torch, dtype), numel, what)

    def _mask(self, mask):
        """reset mask [n]: uint8 or bool (the same byte layout; step_rows hands `done` out as bool)."""
        import torch
        if mask is not None and mask.dtype not in (torch.uint8, torch.bool):
            raise TrexError(E_INVALID, "mask: expected dtype uint8 or bool, got %s" % mask.dtype)
        return _ptr(mask, self.device, None, self.num_envs, "mask")

    def set_wave_balance(self, mode):
        """-1 auto (on from 2048 envs), 0 off (workgroup k runs env k), 1 on: include/trex_batch.h."""
        check(lib.trex_batch_set_wave_balance(self.h, int(mode)))

    def forget_buffers(self):
        check(lib.trex_batch_forget_buffers(self.h))

    pen_in_rows = False

    def set_penalties_in_rows(self, enabled):
        self.pen_in_rows = bool(enabled)
        """Row-block calls write the three penalties to columns 3J+2 .. 3J+4 (row_stride >= 3J + 5): include/trex_batch.h."""
        check(lib.trex_batch_set_penalties_in_rows(self.h, 1 if enabled else 0))

    def reset(self, obs_out=None, mask=None, stream=None):
        n, J = self.num_envs, self.J
        check(lib.trex_batch_reset(self.h, self._mask(mask), self._p(obs_out, "float32", n * 3 * J, "obs_out"),
                                   self._stream(stream)))

    def step(self, actions, obs, reward, done, penalties=None, stream=None):
        n, J = self.num_envs, self.J
        check(lib.trex_batch_step(self.h, self._p(actions, "float32", n * J, "actions"), self._p(obs, "float32", n * 3 * J, "obs"),
                                  self._p(reward, "float32", n, "reward"), self._p(done, "uint8", n, "done"),
                                  self._p(penalties, "float32", 3 * n, "penalties"), self._stream(stream)))

    def step_rows(self, actions, rows, penalties=None, stream=None, done=None):
        """One step writing the [n, stride] row block obs | reward | done (stride = rows.shape[1] >= 3J + 2);
        done [n] uint8 or bool, optional: the flags once more as bytes."""
        import torch
        n, J = self.num_envs, self.J
        if done is not None and done.dtype not in (torch.uint8, torch.bool):
            raise TrexError(E_INVALID, "done: expected dtype uint8 or bool, got %s" % done.dtype)
        check(lib.trex_batch_step_rows(self.h, self._p(actions, "float32", n * J, "actions"),
                                       self._p(rows, "float32", (n - 1) * rows.shape[1] + 3 * J + (5 if self.pen_in_rows else 2), "rows"), int(rows.shape[1]),
                                       self._p(penalties, "float32", 3 * n, "penalties"),
                                       _ptr(done, self.device, None, n, "done"), self._stream(stream)))

    def step_many(self, actions, rows, penalties=None, done=None, stream=None):
        """S env-steps in one launch: actions [S, n, J], rows [S, n, stride] (obs | reward | done per row)."""
        import torch
        n, J, S = self.num_envs, self.J, int(actions.shape[0])
        if done is not None and done.dtype not in (torch.uint8, torch.bool):
            raise TrexError(E_INVALID, "done: expected dtype uint8 or bool, got %s" % done.dtype)
        if rows.dim() != 3 or int(rows.shape[0]) != S or int(rows.shape[1]) != n:
            raise TrexError(E_INVALID, "rows: expected shape [%d, %d, >= %d], got %s" % (S, n, 3 * J + 2, tuple(rows.shape)))
        check(lib.trex_batch_step_many(self.h, self._p(actions, "float32", S * n * J, "actions"),
                                       self._p(rows, "float32", (S * n - 1) * rows.shape[2] + 3 * J + (5 if self.pen_in_rows else 2), "rows"), int(rows.shape[2]), S,
                                       self._p(penalties, "float32", 3 * S * n, "penalties"), _ptr(done, self.device, None, S * n, "done"),
                                       self._stream(stream)))

    def reset_rows(self, rows, mask=None, stream=None):
        n, J = self.num_envs, self.J
        check(lib.trex_batch_reset_rows(self.h, self._mask(mask),
                                        self._p(rows, "float32", (n - 1) * rows.shape[1] + 3 * J + (5 if self.pen_in_rows else 2), "rows"),
                                        int(rows.shape[1]), self._stream(stream)))

    def set_episode_limit(self, max_episode_steps, episode_steps=None, stream=None):
        """Episode limit inside the step launch (0 = off); episode_steps [n] int32 sets the per-env counts."""
        check(lib.trex_batch_set_episode_limit(self.h, int(max_episode_steps), self._p(episode_steps, "int32", self.num_envs, "episode_steps"),
                                               self._stream(stream)))

    def get_episode_steps(self, out, stream=None):
        check(lib.trex_batch_get_episode_steps(self.h, self._p(out, "int32", self.num_envs, "episode_steps"), self._stream(stream)))

    def debug_step(self, actions, obs, debug, stream=None):
        n, J = self.num_envs, self.J
        check(lib.trex_batch_debug_step(self.h, self._p(actions, "float32", n * J, "actions"),
                                        self._p(obs, "float32", n * 3 * J, "obs"), self._p(debug, "float32", 4096, "debug"),
                                        self._stream(stream)))

    def get_state(self, out, stream=None):
        check(lib.trex_batch_get_state(self.h, self._p(out, "float32", self.num_envs * self.state_width, "state"),
                                       self._stream(stream)))

    def set_state(self, state, stream=None):
        check(lib.trex_batch_set_state(self.h, self._p(state, "float32", self.num_envs * self.state_width, "state"),
                                       self._stream(stream)))

    def set_motors_enabled(self, enabled, stream=None):
        check(lib.trex_batch_set_motors_enabled(self.h, int(enabled), self._stream(stream)))

    def head_position(self, out, stream=None):
        check(lib.trex_batch_head_position(self.h, self._p(out, "float32", 3 * self.num_envs, "out"), self._stream(stream)))

    def link_transforms(self, out, stream=None):
        check(lib.trex_batch_link_transforms(self.h, self._p(out, "float32", None, "out"), self._stream(stream)))

    def visual_transforms(self, out, stream=None):
        check(lib.trex_batch_visual_transforms(self.h, self._p(out, "float32", None, "out"), self._stream(stream)))

    def set_domain(self, mass_scale=None, friction=None, stream=None):
        n = self.num_envs
        check(lib.trex_batch_set_domain(self.h, self._p(mass_scale, "float32", n * self.model.num_bodies, "mass_scale"),
                                        self._p(friction, "float32", n, "friction"), self._stream(stream)))

    def contact_stats(self, count=None, normal_impulse=None, stream=None):
        n = self.num_envs
        check(lib.trex_batch_contact_stats(self.h, self._p(count, "int32", n, "count"),
                                           self._p(normal_impulse, "float32", n, "normal_impulse"), self._stream(stream)))

    def launch_info(self):
        g, b, l, a = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(lib.trex_batch_launch_info(self.h, C.byref(g), C.byref(b), C.byref(l), C.byref(a)))
        return dict(grid=g.value, block=b.value, lds_bytes=l.value, alg_bytes_per_env_step=a.value)

    def time_steps(self, actions, obs, reward, done, steps, stream=None):
        ms = C.c_float()
        n, J = self.num_envs, self.J
        check(lib.trex_batch_time_steps(self.h, self._p(actions, "float32", n * J, "actions"), self._p(obs, "float32", n * 3 * J, "obs"),
                                        self._p(reward, "float32", n, "reward"), self._p(done, "uint8", n, "done"), int(steps),
                                        self._stream(stream), C.byref(ms)))
        return ms.value


class Policy:
    """Trainer-side kernels of include/trex_policy.h for N envs on one GPU: VecNormalize state + MlpPolicy.step +
    GAE + Adam. All tensors are torch HIP tensors owned by the caller; the parameter vector is ONE flat f32 tensor."""

    PARAM_NAMES = ["pi.W1", "pi.b1", "pi.W2", "pi.b2", "pi.W3", "pi.b3", "vf.W1", "vf.b1", "vf.W2", "vf.b2", "vf.W3", "vf.b3",
                   "logstd"]

    def __init__(self, num_envs, obs_dim, act_dim, hidden=64, device=0):
        self.n, self.D, self.A, self.H, self.device = int(num_envs), int(obs_dim), int(act_dim), int(hidden), int(device)
        h = _vp()
        check(lib.trex_policy_create(self.n, self.D, self.A, self.H, self.device, C.byref(h)))
        self.h = h
        self.param_count = lib.trex_policy_param_count(h)
        off = (C.c_int * 13)()
        check(lib.trex_policy_param_offsets(h, off))
        D, A, H = self.D, self.A, self.H
        shapes = [(D, H), (H,), (H, H), (H,), (H, A), (A,), (D, H), (H,), (H, H), (H,), (H, 1), (1,), (A,)]
        self.layout = {n: (int(o), sh) for n, o, sh in zip(self.PARAM_NAMES, off, shapes)}

    def close(self):
        if getattr(self, "h", None):
            lib.trex_policy_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _p(self, t, numel, what, dtype="float32"):
        import torch
        return _ptr(t, self.device, getattr(torch, dtype), numel, what)

    _stream = staticmethod(Batch._stream)

    def get_stats(self):
        """dict of VecNormalize's running statistics (host, f64)."""
        D = self.D
        buf = (C.c_double * (2 * D + 5))()
        check(lib.trex_policy_get_stats(self.h, buf, self._stream(None)))
        a = np.array(buf[:])
        return dict(obs_mean=a[:D], obs_var=a[D:2 * D], obs_count=a[2 * D], ret_mean=a[2 * D + 1], ret_var=a[2 * D + 2],
                    ret_count=a[2 * D + 3], raw_reward_sum=a[2 * D + 4])

    def set_stats(self, st):
        D = self.D
        a = np.concatenate([np.asarray(st["obs_mean"], float).reshape(D), np.asarray(st["obs_var"], float).reshape(D),
                            [st["obs_count"], st["ret_mean"], st["ret_var"], st["ret_count"], st.get("raw_reward_sum", 0.0)]])
        check(lib.trex_policy_set_stats(self.h, (C.c_double * (2 * D + 5))(*a), self._stream(None)))

    def get_returns(self, out):
        check(lib.trex_policy_get_returns(self.h, self._p(out, self.n, "ret"), self._stream(None)))

    def observe(self, rows, with_reward=True, gamma=0.99, raw_rew_out=None, done_out=None, rew_scale_out=None, stream=None):
        n = self.n
        check(lib.trex_policy_observe(self.h, self._p(rows, (n - 1) * rows.shape[1] + self.D + (2 if with_reward else 0), "rows"),
                                      int(rows.shape[1]), int(bool(with_reward)), float(gamma), self._p(raw_rew_out, n, "raw_rew_out"),
                                      self._p(done_out, n, "done_out"), self._p(rew_scale_out, 1, "rew_scale_out"),
                                      self._stream(stream)))

    def act(self, theta, rows, noise, actions, obs_out=None, act_out=None, logp_out=None, value_out=None, clip_obs=10.0,
            value_only=False, stream=None):
        n, D, A = self.n, self.D, self.A
        check(lib.trex_policy_act(self.h, self._p(theta, self.param_count, "theta"), self._p(rows, (n - 1) * rows.shape[1] + D, "rows"),
                                  int(rows.shape[1]), float(clip_obs), self._p(noise, n * A, "noise"), self._p(actions, n * A, "actions"),
                                  self._p(obs_out, n * D, "obs_out"), self._p(act_out, n * A, "act_out"),
                                  self._p(logp_out, n, "logp_out"), self._p(value_out, n, "value_out"), int(bool(value_only)),
                                  self._stream(stream)))

    def gae(self, raw_rew, rew_scale, done, values, adv, ret, gamma=0.99, lam=0.95, clip_rew=10.0, stream=None):
        T, n = int(raw_rew.shape[0]), self.n
        check(lib.trex_policy_gae(self.h, self._p(raw_rew, T * n, "raw_rew"), self._p(rew_scale, T, "rew_scale"),
                                  self._p(done, T * n, "done"), self._p(values, (T + 1) * n, "values"), self._p(adv, T * n, "adv"),
                                  self._p(ret, T * n, "ret"), T, float(gamma), float(lam), float(clip_rew), self._stream(stream)))

    def adam(self, theta, grad, m, v, lr=3e-4, beta1=0.9, beta2=0.999, eps=1e-5, max_grad_norm=0.5, grad_norm_out=None, stream=None):
        P = self.param_count
        check(lib.trex_policy_adam(self.h, self._p(theta, P, "theta"), self._p(grad, P, "grad"), self._p(m, P, "m"), self._p(v, P, "v"),
                                   float(lr), float(beta1), float(beta2), float(eps), float(max_grad_norm),
                                   self._p(grad_norm_out, 1, "grad_norm_out"), self._stream(stream)))

    def adam_reset(self, stream=None):
        check(lib.trex_policy_adam_reset(self.h, self._stream(stream)))

    def minibatch_stats(self, adv, perm, num_minibatches, mb, out, stream=None):
        N = adv.numel()
        check(lib.trex_policy_minibatch_stats(self.h, self._p(adv, N, "adv"), N, self._p(perm, num_minibatches * mb, "perm", "int64"),
                                              int(num_minibatches), int(mb), self._p(out, 2 * num_minibatches, "stats_out"),
                                              self._stream(stream)))

    def minibatch_grad(self, theta, grad, obs, act, logp, val, adv, ret, perm, first, mb, adv_stats, cliprange=0.2, ent_coef=0.0,
                       vf_coef=0.5, grad_scale=1.0, loss_sums=None, stream=None):
        """The gradient of perm[first : first + mb] alone, times grad_scale (no clip, no Adam): the data-parallel trainer's
        half of a minibatch step (include/trex_policy.h)."""
        P, N, D, A = self.param_count, adv.numel(), self.D, self.A
        check(lib.trex_policy_minibatch_grad(
            self.h, self._p(theta, P, "theta"), self._p(grad, P, "grad"), self._p(obs, N * D, "obs"), self._p(act, N * A, "act"),
            self._p(logp, N, "logp"), self._p(val, N, "val"), self._p(adv, N, "adv"), self._p(ret, N, "ret"), N,
            self._p(perm, first + mb, "perm", "int64"), int(first), int(mb), self._p(adv_stats, 2, "adv_stats"), float(cliprange),
            float(ent_coef), float(vf_coef), float(grad_scale), self._p(loss_sums, 2, "loss_sums"), self._stream(stream)))

    def minibatch_step(self, theta, grad, m, v, obs, act, logp, val, adv, ret, perm, first, mb, adv_stats, cliprange=0.2,
                       ent_coef=0.0, vf_coef=0.5, lr=3e-4, beta1=0.9, beta2=0.999, eps=1e-5, max_grad_norm=0.5, loss_sums=None,
                       stream=None):
        """One PPO2 minibatch step on perm[first : first + mb] (include/trex_policy.h): three launches."""
        P, N, D, A = self.param_count, adv.numel(), self.D, self.A
        check(lib.trex_policy_minibatch_step(
            self.h, self._p(theta, P, "theta"), self._p(grad, P, "grad"), self._p(m, P, "m"), self._p(v, P, "v"),
            self._p(obs, N * D, "obs"), self._p(act, N * A, "act"), self._p(logp, N, "logp"), self._p(val, N, "val"),
            self._p(adv, N, "adv"), self._p(ret, N, "ret"), N, self._p(perm, first + mb, "perm", "int64"), int(first), int(mb),
            self._p(adv_stats, 2, "adv_stats"), float(cliprange), float(ent_coef), float(vf_coef), float(lr), float(beta1),
            float(beta2), float(eps), float(max_grad_norm), self._p(loss_sums, 2, "loss_sums"), self._stream(stream)))
