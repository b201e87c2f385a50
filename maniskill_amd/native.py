"""ctypes binding of the C ABI in include/mssim.h.

`NativeLib(path, prefix)` binds any library that exports the ABI under a symbol prefix. The
product library is `maniskill_amd/_native/libmssim.so` (prefix `mssim_`, HIP/gfx950); loading it
fails loudly if it has not been built -- there is no CPU fallback in this package.
"""
import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

from . import PACKAGE_DIR
from .model.compile import CompiledModel

ABI_VERSION = 6
NATIVE_LIB_PATH = os.environ.get("MSSIM_LIB") or os.path.join(PACKAGE_DIR, "_native", "libmssim.so")  # MSSIM_LIB: debug builds of the same HIP library

# apply / fetch selector bits (include/mssim.h)
RIGID_DATA = 1 << 0
ART_QPOS = 1 << 1
ART_QVEL = 1 << 2
ART_QF = 1 << 3
ART_ROOT_POSE = 1 << 4
ART_ROOT_VEL = 1 << 5
ART_TARGET_POS = 1 << 6
ART_TARGET_VEL = 1 << 7
RIGID_FORCE = 1 << 8
LINK_POSE = 1 << 9
LINK_VEL = 1 << 10
ART_QACC = 1 << 11
ALL = 0xFFF

_I32P = C.POINTER(C.c_int32)
_F32P = C.POINTER(C.c_float)


class ModelDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("n_dof", C.c_int32),
        ("dof_parent", _I32P),
        ("dof_type", _I32P),
        ("dof_frame", _F32P),
        ("dof_axis", _F32P),
        ("dof_limit", _F32P),
        ("dof_drive", _F32P),
        ("dof_armature", _F32P),
        ("body_inertial", _F32P),
        ("body_gravity", _I32P),
        ("n_tendon", C.c_int32),
        ("tendon_dof", _I32P),
        ("tendon_param", _F32P),
        ("n_link", C.c_int32),
        ("link_body", _I32P),
        ("link_frame", _F32P),
        ("n_free", C.c_int32),
        ("free_inertial", _F32P),
        ("free_damping", _F32P),
        ("free_gravity", _I32P),
        ("n_kin", C.c_int32),
        ("n_shape", C.c_int32),
        ("shape_type", _I32P),
        ("shape_body_kind", _I32P),
        ("shape_body_index", _I32P),
        ("shape_row", _I32P),
        ("shape_frame", _F32P),
        ("shape_param", _F32P),
        ("shape_material", _F32P),
        ("shape_hull", _I32P),
        ("shape_bound", _F32P),
        ("n_hull_verts", C.c_int32),
        ("hull_verts", _F32P),
        ("n_pair", C.c_int32),
        ("pair_shape", _I32P),
        ("gravity", C.c_float * 3),
        ("timestep", C.c_float),
        ("contact_offset", C.c_float),
        ("rest_offset", C.c_float),
        ("bounce_threshold", C.c_float),
        ("position_iterations", C.c_int32),
        ("velocity_iterations", C.c_int32),
        ("erp", C.c_float),
        ("max_depenetration_velocity", C.c_float),
        ("sleep_threshold", C.c_float),
        ("num_envs", C.c_int32),
        ("n_env_shape", C.c_int32),
        ("shape_env_slot", _I32P),
        ("env_shape_frame", _F32P),
        ("env_shape_param", _F32P),
        ("env_shape_bound", _F32P),
        ("n_env_free", C.c_int32),
        ("free_env_slot", _I32P),
        ("env_free_inertial", _F32P),
        ("n_tri", C.c_int32),
        ("tri_soup", _F32P),
        ("n_tri_node", C.c_int32),
        ("tri_bvh", _F32P),
    ]


class Buffers(C.Structure):
    _fields_ = [
        ("rigid_body_data", C.c_void_p),
        ("rigid_body_force", C.c_void_p),
        ("art_qpos", C.c_void_p),
        ("art_qvel", C.c_void_p),
        ("art_qacc", C.c_void_p),
        ("art_qf", C.c_void_p),
        ("art_target_qpos", C.c_void_p),
        ("art_target_qvel", C.c_void_p),
    ]


def make_model_desc(model: CompiledModel):
    """Returns (ModelDesc, keepalive) -- keepalive must outlive the create call."""
    d = ModelDesc()
    keep = []
    d.abi_version = ABI_VERSION
    for name, ctype in ModelDesc._fields_:
        if ctype in (_I32P, _F32P):
            a = model.arrays[name]
            want = np.int32 if ctype is _I32P else np.float32
            a = np.ascontiguousarray(a, dtype=want)
            keep.append(a)
            setattr(d, name, a.ctypes.data_as(ctype))
        elif name == "gravity":
            d.gravity = (C.c_float * 3)(*model.scalars["gravity"])
        elif name != "abi_version":
            setattr(d, name, model.scalars[name])
    return d, keep


class PickTask(C.Structure):
    _fields_ = [
        ("tcp_row", C.c_int32), ("obj_row", C.c_int32), ("goal_row", C.c_int32), ("finger1_row", C.c_int32), ("finger2_row", C.c_int32),
        ("n_static_dofs", C.c_int32), ("goal_thresh", C.c_float), ("static_thresh", C.c_float), ("min_force", C.c_float),
        ("max_angle_deg", C.c_float), ("reward_scale", C.c_float), ("elapsed_steps", C.c_void_p), ("elapsed_out", C.c_void_p), ("truncated_out", C.c_void_p), ("time_limit", C.c_int32), ("terminated_out", C.c_void_p),
    ]


class PushTask(C.Structure):
    _fields_ = [("tcp_row", C.c_int32), ("obj_row", C.c_int32), ("goal_row", C.c_int32), ("goal_radius", C.c_float),
                ("cube_half_size", C.c_float), ("reward_scale", C.c_float), ("elapsed_steps", C.c_void_p), ("elapsed_out", C.c_void_p), ("truncated_out", C.c_void_p), ("time_limit", C.c_int32), ("terminated_out", C.c_void_p)]


class PegTask(C.Structure):
    _fields_ = [("tcp_row", C.c_int32), ("peg_row", C.c_int32), ("box_row", C.c_int32), ("finger1_row", C.c_int32), ("finger2_row", C.c_int32),
                ("min_force", C.c_float), ("max_angle_deg", C.c_float), ("reward_scale", C.c_float),
                ("peg_half_sizes", C.c_void_p), ("box_hole_offsets", C.c_void_p), ("box_hole_radii", C.c_void_p),
                ("elapsed_steps", C.c_void_p), ("elapsed_out", C.c_void_p), ("truncated_out", C.c_void_p), ("time_limit", C.c_int32), ("terminated_out", C.c_void_p)]


class NativeError(RuntimeError):
    pass


class NativeLib:
    """One loaded library exporting the mssim ABI under `prefix`."""

    _cache: Dict[tuple, "NativeLib"] = {}

    def __init__(self, path: str, prefix: str = "mssim_"):
        if not os.path.exists(path):
            raise NativeError(
                f"native simulation library not found: {path}. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback."
            )
        self.path, self.prefix = path, prefix
        self.lib = C.CDLL(path)
        f = self._fn
        H = C.c_void_p
        f("create", C.c_int, [C.POINTER(ModelDesc), C.c_int32, C.c_int32, C.POINTER(H)])
        f("destroy", None, [H])
        f("bind_buffers", C.c_int, [H, C.POINTER(Buffers)])
        f("set_timestep", C.c_int, [H, C.c_float])
        f("get_timestep", C.c_float, [H])
        f("apply", C.c_int, [H, C.c_uint32, C.c_void_p])
        f("fetch", C.c_int, [H, C.c_uint32, C.c_void_p])
        f("step", C.c_int, [H, C.c_int32, C.c_void_p])
        f("update_kinematics", C.c_int, [H, C.c_void_p])
        f("wake_all", C.c_int, [H, C.c_void_p])
        f("wake_envs", C.c_int, [H, C.c_void_p, C.c_int32, C.c_void_p])
        f("create_pair_query", C.c_int, [H, _I32P, C.c_int32, _I32P])
        f("query_pair_impulses", C.c_int, [H, C.c_int32, C.c_void_p, C.c_void_p])
        f("create_body_query", C.c_int, [H, _I32P, C.c_int32, _I32P])
        f("query_body_impulses", C.c_int, [H, C.c_int32, C.c_void_p, C.c_void_p])
        f("set_drive_properties", C.c_int, [H, _F32P])
        f("read_internal", C.c_int, [H, C.c_char_p, C.c_void_p, C.c_int32, C.c_void_p])
        f("link_jacobian", C.c_int, [H, C.c_int32, C.c_void_p, C.c_void_p])
        f("overflow_count", C.c_int, [H, C.c_void_p])
        f("set_action_map", C.c_int, [H, _I32P, _F32P, _F32P, _I32P])
        f("set_ee_action_map", C.c_int, [H, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_int32])
        f("apply_action", C.c_int, [H, C.c_void_p, C.c_int32, C.c_void_p])
        f("step_action", C.c_int, [H, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p])
        f("defer_fetch", C.c_int, [H, C.c_uint32])
        f("defer_step_action", C.c_int, [H, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p])
        f("task_pick_outputs", C.c_int, [H, C.POINTER(PickTask), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p])
        f("task_peg_outputs", C.c_int, [H, C.POINTER(PegTask), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p])
        f("task_push_outputs", C.c_int, [H, C.POINTER(PushTask), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p])
        f("profile_enable", C.c_int, [H, C.c_int32])
        f("profile_read", C.c_int, [H, _F32P, _I32P])
        f("last_error", C.c_char_p, [H])
        f("abi_version", C.c_int, [])
        if self.abi_version() != ABI_VERSION:
            raise NativeError(f"{path}: ABI version {self.abi_version()} != {ABI_VERSION}")

    EXPORTS = [
        "create", "destroy", "bind_buffers", "set_timestep", "get_timestep", "apply", "fetch", "step",
        "update_kinematics", "wake_all", "wake_envs", "create_pair_query", "query_pair_impulses", "create_body_query",
        "query_body_impulses", "set_drive_properties", "read_internal", "link_jacobian", "overflow_count", "set_action_map", "set_ee_action_map",
        "apply_action", "step_action", "defer_fetch", "defer_step_action", "task_pick_outputs", "task_push_outputs", "task_peg_outputs", "profile_enable",
        "profile_read", "last_error",
        "abi_version",
    ]

    def _fn(self, name, restype, argtypes):
        fn = getattr(self.lib, self.prefix + name)
        fn.restype, fn.argtypes = restype, argtypes
        setattr(self, name, fn)

    @classmethod
    def load(cls, path: Optional[str] = None, prefix: str = "mssim_") -> "NativeLib":
        path = path or NATIVE_LIB_PATH
        key = (os.path.abspath(path), prefix)
        if key not in cls._cache:
            cls._cache[key] = cls(path, prefix)
        return cls._cache[key]


class NativeSim:
    """Owns one `mssim_handle`."""

    def __init__(self, lib: NativeLib, model: CompiledModel, num_envs: int, device: int):
        self.lib, self.model, self.num_envs, self.device = lib, model, num_envs, device
        desc, keep = make_model_desc(model)
        h = C.c_void_p()
        rc = lib.create(C.byref(desc), num_envs, device, C.byref(h))
        if rc != 0:
            msg = lib.last_error(None)
            raise NativeError(f"mssim create failed ({rc}): {msg.decode() if msg else ''}")
        self.h = h
        self._queries = {}

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.last_error(self.h)
            raise NativeError(f"mssim {what} failed ({rc}): {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bind(self, **ptrs):
        b = Buffers()
        for k, v in ptrs.items():
            setattr(b, k, v)
        self._check(self.lib.bind_buffers(self.h, C.byref(b)), "bind_buffers")

    def apply(self, what, stream=None):
        self._check(self.lib.apply(self.h, what, stream), "apply")

    def fetch(self, what, stream=None):
        self._check(self.lib.fetch(self.h, what, stream), "fetch")

    def defer_step_action(self, action_ptr, action_dim, n_substeps, stream=None):
        self._check(self.lib.defer_step_action(self.h, action_ptr, action_dim, n_substeps, stream), "defer_step_action")

    def defer_fetch(self, what):
        self._check(self.lib.defer_fetch(self.h, what), "defer_fetch")

    def step(self, n_substeps=1, stream=None):
        self._check(self.lib.step(self.h, n_substeps, stream), "step")

    def wake_all(self, stream=None):
        self._check(self.lib.wake_all(self.h, stream), "wake_all")

    def wake_envs(self, env_idx_ptr, n_idx, stream=None):
        self._check(self.lib.wake_envs(self.h, env_idx_ptr, n_idx, stream), "wake_envs")

    def update_kinematics(self, stream=None):
        self._check(self.lib.update_kinematics(self.h, stream), "update_kinematics")

    def set_timestep(self, dt):
        self._check(self.lib.set_timestep(self.h, dt), "set_timestep")

    def get_timestep(self):
        return float(self.lib.get_timestep(self.h))

    def create_pair_query(self, body_pairs):
        a = np.ascontiguousarray(body_pairs, dtype=np.int32).reshape(-1, 2)
        qid = C.c_int32()
        self._check(self.lib.create_pair_query(self.h, a.ctypes.data_as(_I32P), len(a), C.byref(qid)), "create_pair_query")
        return qid.value

    def query_pair_impulses(self, qid, out_ptr, stream=None):
        self._check(self.lib.query_pair_impulses(self.h, qid, out_ptr, stream), "query_pair_impulses")

    def create_body_query(self, rows):
        a = np.ascontiguousarray(rows, dtype=np.int32).reshape(-1)
        qid = C.c_int32()
        self._check(self.lib.create_body_query(self.h, a.ctypes.data_as(_I32P), len(a), C.byref(qid)), "create_body_query")
        return qid.value

    def query_body_impulses(self, qid, out_ptr, stream=None):
        self._check(self.lib.query_body_impulses(self.h, qid, out_ptr, stream), "query_body_impulses")

    def set_drive_properties(self, drive):
        a = np.ascontiguousarray(drive, dtype=np.float32).reshape(-1, 4)
        assert a.shape[0] == self.model.n_dof
        self._check(self.lib.set_drive_properties(self.h, a.ctypes.data_as(_F32P)), "set_drive_properties")

    def read_internal(self, name, out_ptr, max_items, stream=None):
        n = self.lib.read_internal(self.h, name.encode(), out_ptr, max_items, stream)
        if n < 0:
            self._check(n, f"read_internal({name})")
        return n

    def link_jacobian(self, link_index, out_ptr, stream=None):
        self._check(self.lib.link_jacobian(self.h, int(link_index), out_ptr, stream), "link_jacobian")

    def set_action_map(self, column, low, high, flags):
        col = np.ascontiguousarray(column, dtype=np.int32)
        lo = np.ascontiguousarray(low, dtype=np.float32)
        hi = np.ascontiguousarray(high, dtype=np.float32)
        fl = np.ascontiguousarray(flags, dtype=np.int32)
        assert len(col) == len(lo) == len(hi) == len(fl) == self.model.n_dof
        self._check(self.lib.set_action_map(self.h, col.ctypes.data_as(_I32P), lo.ctypes.data_as(_F32P), hi.ctypes.data_as(_F32P), fl.ctypes.data_as(_I32P)), "set_action_map")

    def set_ee_action_map(self, link_index, column0, rows, low, high, rot_scale, flags):
        self._check(self.lib.set_ee_action_map(self.h, int(link_index), int(column0), int(rows), float(low), float(high), float(rot_scale), int(flags)), "set_ee_action_map")

    def apply_action(self, action_ptr, action_dim, stream=None):
        self._check(self.lib.apply_action(self.h, action_ptr, action_dim, stream), "apply_action")

    def step_action(self, action_ptr, action_dim, n_substeps, stream=None):
        self._check(self.lib.step_action(self.h, action_ptr, action_dim, n_substeps, stream), "step_action")

    def task_peg_outputs(self, task: "PegTask", obs_ptr, reward_ptr, flags_ptr, head_ptr, stream=None):
        self._check(self.lib.task_peg_outputs(self.h, C.byref(task), obs_ptr, reward_ptr, flags_ptr, head_ptr, stream), "task_peg_outputs")

    def task_push_outputs(self, task: "PushTask", obs_ptr, reward_ptr, flags_ptr, stream=None):
        self._check(self.lib.task_push_outputs(self.h, C.byref(task), obs_ptr, reward_ptr, flags_ptr, stream), "task_push_outputs")

    def task_pick_outputs(self, task: "PickTask", obs_ptr, reward_ptr, flags_ptr, stream=None):
        self._check(self.lib.task_pick_outputs(self.h, C.byref(task), obs_ptr, reward_ptr, flags_ptr, stream), "task_pick_outputs")

    def profile_enable(self, on=True):
        self._check(self.lib.profile_enable(self.h, 1 if on else 0), "profile_enable")

    def profile_read(self):
        """-> {"solve": (ms, launches), "narrow": (ms, launches)} since the last read"""
        ms = (C.c_float * 2)()
        cnt = (C.c_int32 * 2)()
        self._check(self.lib.profile_read(self.h, ms, cnt), "profile_read")
        return {"solve": (float(ms[0]), int(cnt[0])), "narrow": (float(ms[1]), int(cnt[1]))}

    def overflow_count(self, stream=None):
        return int(self.lib.overflow_count(self.h, stream))
