"""`px`-shaped system object over the native core.

Counterpart of `sapien.physx.PhysxGpuSystem` as ManiSkill uses it through `ManiSkillScene.px`
(mani_skill/envs/scene.py:60-68; complete method list in SURVEY.md 8b): `gpu_init`, `step`,
`timestep`, `gpu_apply_*`, `gpu_fetch_*`, `gpu_update_articulation_kinematics`, `cuda_*` buffers
with `.torch()`, contact impulse queries.

The default backend is the HIP library (`maniskill_amd/_native/libmssim.so`) on a `cuda:*`
(ROCm) device and raises if it is missing; other backends only exist if a caller registers them
(`register_backend`), which the package itself never does.
"""
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import native
from ..model.compile import CompiledModel


class CudaArray:
    """Mimics SAPIEN's `CudaArray`: `.torch()` returns the aliased tensor (structs/base.py:112-114)."""

    def __init__(self, tensor: torch.Tensor):
        self._t = tensor

    def torch(self) -> torch.Tensor:
        return self._t

    @property
    def shape(self):
        return tuple(self._t.shape)

    @property
    def ptr(self):
        return self._t.data_ptr()


class ContactPairImpulseQuery:
    def __init__(self, qid: int, impulses: torch.Tensor):
        self.id = qid
        self.cuda_impulses = CudaArray(impulses)


class ContactBodyImpulseQuery(ContactPairImpulseQuery):
    pass


# backend name -> factory(device) -> (NativeLib, native_device_ordinal)
_BACKENDS: Dict[str, Callable] = {}


def register_backend(name: str, factory: Callable):
    """Register an alternative native library for a sim backend name (used by tests to plug the
    CPU oracle in as `physx_cpu`; the package registers nothing but the HIP library)."""
    _BACKENDS[name] = factory


def _hip_backend(device: torch.device):
    if device.type != "cuda":
        raise RuntimeError(
            f"maniskill_amd's simulation backend is HIP-only: device {device} is not a ROCm GPU. "
            "There is no CPU fallback (register a backend explicitly for tests)."
        )
    if not torch.cuda.is_available():
        raise RuntimeError("no ROCm GPU visible: the HIP simulation core cannot run")
    lib = native.NativeLib.load()  # raises NativeError when libmssim.so is not built
    return lib, (device.index if device.index is not None else torch.cuda.current_device())


_BACKENDS["physx_cuda"] = _hip_backend


class MssimSystem:
    """Batched rigid-body system for `num_envs` identical environments."""

    def __init__(self, device="cuda", backend: str = "physx_cuda"):
        self.device = torch.device(device)
        self.backend = backend
        if backend not in _BACKENDS:
            raise RuntimeError(
                f"sim backend {backend!r} is not available; only the HIP backend 'physx_cuda' ships with this package"
            )
        self._lib, self._dev_ordinal = _BACKENDS[backend](self.device)
        self._timestep = 0.01
        self.model: Optional[CompiledModel] = None
        self.num_envs = 0
        self._sim: Optional[native.NativeSim] = None
        self._queries: List = []

    # ------------------------------------------------------------------ lifecycle
    @property
    def timestep(self) -> float:
        return self._timestep

    @timestep.setter
    def timestep(self, dt: float):
        self._timestep = float(dt)
        if self._sim is not None:
            self._sim.set_timestep(self._timestep)

    @property
    def is_initialized(self):
        return self._sim is not None

    def _stream(self):
        if self.device.type == "cuda":
            return torch.cuda.current_stream(self.device).cuda_stream
        return None

    def gpu_init(self, model: CompiledModel, num_envs: int):
        """px.gpu_init() (scene.py:905): compile-time tables are uploaded, state buffers allocated
        and filled with the builders' initial poses."""
        assert self._sim is None, "gpu_init called twice"
        self.model, self.num_envs = model, int(num_envs)
        model.scalars["timestep"] = self._timestep
        if model.scalars.get("n_env_shape", 0) or model.scalars.get("n_env_free", 0):
            assert model.scalars["num_envs"] == self.num_envs, "model with per-env geometry was compiled for a different num_envs"
        else:
            model.scalars["num_envs"] = self.num_envs
        N, R, n = self.num_envs, model.n_rows, model.n_dof
        dev = self.device
        f = dict(dtype=torch.float32, device=dev)
        self._rigid = torch.zeros((max(R, 1) * N, 13), **f)
        self._rigid[:, 3] = 1.0
        self._force = torch.zeros((max(R, 1) * N, 4), **f)
        nd = max(n, 1)
        self._qpos = torch.zeros((N, nd), **f)
        self._qvel = torch.zeros((N, nd), **f)
        self._qacc = torch.zeros((N, nd), **f)
        self._qf = torch.zeros((N, nd), **f)
        self._tqpos = torch.zeros((N, nd), **f)
        self._tqvel = torch.zeros((N, nd), **f)
        self.cuda_rigid_body_data = CudaArray(self._rigid)
        self.cuda_rigid_body_force = CudaArray(self._force)
        self.cuda_articulation_qpos = CudaArray(self._qpos)
        self.cuda_articulation_qvel = CudaArray(self._qvel)
        self.cuda_articulation_qacc = CudaArray(self._qacc)
        self.cuda_articulation_qf = CudaArray(self._qf)
        self.cuda_articulation_target_qpos = CudaArray(self._tqpos)
        self.cuda_articulation_target_qvel = CudaArray(self._tqvel)
        if dev.type == "cuda":
            with torch.cuda.device(dev):
                self._sim = native.NativeSim(self._lib, model, N, self._dev_ordinal)
        else:
            self._sim = native.NativeSim(self._lib, model, N, -1)
        self._sim.bind(
            rigid_body_data=self._rigid.data_ptr(),
            rigid_body_force=self._force.data_ptr(),
            art_qpos=self._qpos.data_ptr(),
            art_qvel=self._qvel.data_ptr(),
            art_qacc=self._qacc.data_ptr(),
            art_qf=self._qf.data_ptr(),
            art_target_qpos=self._tqpos.data_ptr(),
            art_target_qvel=self._tqvel.data_ptr(),
        )
        # initial poses from the builders
        A = model.arrays
        if model.n_link > 0:
            self._rigid[0:N, :7] = torch.from_numpy(A["init_root_pose"]).to(dev)
        for b in range(model.n_free):
            r = model.n_link + b
            self._rigid[r * N : (r + 1) * N, :7] = torch.from_numpy(A["init_free_pose"][b]).to(dev)
        for k in range(model.n_kin):
            r = model.n_link + model.n_free + k
            self._rigid[r * N : (r + 1) * N, :7] = torch.from_numpy(A["init_kin_pose"][k]).to(dev)
        self._sim.apply(native.ALL, self._stream())
        self._sim.update_kinematics(self._stream())
        self._sim.fetch(native.ALL, self._stream())

    def close(self):
        if self._sim is not None:
            self._sim.close()
            self._sim = None

    # ------------------------------------------------------------------ stepping
    def step(self, n_substeps: int = 1):
        """px.step() (scene.py:374-375). `n_substeps > 1` fuses the substep loop of
        BaseEnv._step_action (sapien_env.py:1016-1021) into one native call."""
        self._sim.step(n_substeps, self._stream())

    def gpu_update_articulation_kinematics(self):
        self._sim.update_kinematics(self._stream())

    def wake_all(self):
        """wake every sleeping free body (their sleep counters are state the buffers do not carry; include/mssim.h)"""
        self._sim.wake_all(self._stream())

    def wake_envs(self, env_idx: torch.Tensor):
        """the same for the listed envs (an int64 index tensor on the system's device): what a reset of those envs does to the
        hidden state (include/mssim.h `wake_envs`); no host synchronisation"""
        idx = env_idx.to(device=self.device, dtype=torch.int64).contiguous()
        self._sim.wake_envs(idx.data_ptr(), int(idx.numel()), self._stream())

    # ------------------------------------------------------------------ apply / fetch
    def _apply(self, what):
        self._sim.apply(what, self._stream())

    def _fetch(self, what):
        self._sim.fetch(what, self._stream())

    def gpu_apply_rigid_dynamic_data(self):
        self._apply(native.RIGID_DATA)

    def gpu_apply_articulation_qpos(self):
        self._apply(native.ART_QPOS)

    def gpu_apply_articulation_qvel(self):
        self._apply(native.ART_QVEL)

    def gpu_apply_articulation_qf(self):
        self._apply(native.ART_QF)

    def gpu_apply_articulation_root_pose(self):
        self._apply(native.ART_ROOT_POSE)

    def gpu_apply_articulation_root_velocity(self):
        self._apply(native.ART_ROOT_VEL)

    def gpu_apply_articulation_target_position(self):
        self._apply(native.ART_TARGET_POS)

    def gpu_apply_articulation_target_velocity(self):
        self._apply(native.ART_TARGET_VEL)

    def gpu_apply_rigid_dynamic_force(self):
        self._apply(native.RIGID_FORCE)

    def gpu_apply_all(self):
        self._apply(native.ALL)

    def gpu_fetch_rigid_dynamic_data(self):
        self._fetch(native.RIGID_DATA)

    def gpu_fetch_articulation_link_pose(self):
        self._fetch(native.LINK_POSE)

    def gpu_fetch_articulation_link_velocity(self):
        self._fetch(native.LINK_VEL)

    def gpu_fetch_articulation_qpos(self):
        self._fetch(native.ART_QPOS)

    def gpu_fetch_articulation_qvel(self):
        self._fetch(native.ART_QVEL)

    def gpu_fetch_articulation_qacc(self):
        self._fetch(native.ART_QACC)

    def gpu_fetch_articulation_target_qpos(self):
        self._fetch(native.ART_TARGET_POS)

    def gpu_fetch_articulation_target_qvel(self):
        self._fetch(native.ART_TARGET_VEL)

    def gpu_fetch_all(self):
        self._fetch(native.ALL)

    def defer_fetch_all(self):
        """gpu_fetch_all owed to the next native call: the task epilogue that follows performs it inside its own
        launch (include/mssim.h `defer_fetch`). Only for callers that issue that call next."""
        self._sim.defer_fetch(native.ALL)

    # ------------------------------------------------------------------ contact queries
    def gpu_create_contact_pair_impulse_query(self, body_pairs: Sequence[Tuple[int, int]]) -> ContactPairImpulseQuery:
        """body_pairs: (row_a, row_b) body rows of `cuda_rigid_body_data` (row index // num_envs),
        -1 for the static world (scene.py:769-772)."""
        pairs = np.asarray(body_pairs, dtype=np.int32).reshape(-1, 2)
        qid = self._sim.create_pair_query(pairs)
        out = torch.zeros((len(pairs) * self.num_envs, 3), dtype=torch.float32, device=self.device)
        q = ContactPairImpulseQuery(qid, out)
        self._queries.append(q)
        return q

    def gpu_query_contact_pair_impulses(self, query: ContactPairImpulseQuery):
        self._sim.query_pair_impulses(query.id, query.cuda_impulses.ptr, self._stream())

    def gpu_create_contact_body_impulse_query(self, body_rows: Sequence[int]) -> ContactBodyImpulseQuery:
        rows = np.asarray(body_rows, dtype=np.int32).reshape(-1)
        qid = self._sim.create_body_query(rows)
        out = torch.zeros((len(rows) * self.num_envs, 3), dtype=torch.float32, device=self.device)
        q = ContactBodyImpulseQuery(qid, out)
        self._queries.append(q)
        return q

    def gpu_query_contact_body_impulses(self, query: ContactBodyImpulseQuery):
        self._sim.query_body_impulses(query.id, query.cuda_impulses.ptr, self._stream())

    # ------------------------------------------------------------------ misc
    def set_drive_properties(self, drive: np.ndarray):
        self.model.arrays["dof_drive"] = np.ascontiguousarray(drive, dtype=np.float32)
        if self._sim is not None:
            self._sim.set_drive_properties(drive)

    def read_internal(self, name: str, max_items: int) -> torch.Tensor:
        out = torch.zeros((max_items, self.num_envs), dtype=torch.float32, device=self.device)
        n = self._sim.read_internal(name, out.data_ptr(), max_items, self._stream())
        return out[:n]

    def link_jacobian(self, link_index: int) -> torch.Tensor:
        """[N, 6, n_dof] geometric Jacobian (linear; angular) of articulation link `link_index` in the root
        frame at the current simulation state (include/mssim.h `link_jacobian`)"""
        out = torch.empty((self.num_envs, 6, self.model.n_dof), dtype=torch.float32, device=self.device)
        self._sim.link_jacobian(link_index, out.data_ptr(), self._stream())
        return out

    # ------------------------------------------------------------------ fused callers (HIP only)
    @property
    def supports_fused_callers(self) -> bool:
        return self.backend == "physx_cuda"

    def set_action_map(self, column, low, high, flags):
        self._sim.set_action_map(column, low, high, flags)

    def set_ee_action_map(self, ee):
        """end-effector block of the action map: `(link_index, column0, rows, low, high, rot_scale, flags)` or None"""
        if ee is None:
            self._sim.set_ee_action_map(-1, 0, 3, 0.0, 0.0, 0.0, 0)
        else:
            self._sim.set_ee_action_map(*ee)

    def apply_action(self, action: torch.Tensor):
        """affine action -> drive targets in one launch (include/mssim.h `apply_action`)"""
        assert action.dtype == torch.float32 and action.is_contiguous() and action.shape[0] == self.num_envs
        self._sim.apply_action(action.data_ptr(), action.shape[1], self._stream())

    def step_action(self, action: torch.Tensor, n_substeps: int, defer: bool = False):
        """`apply_action` + `step(n_substeps)` in one launch (include/mssim.h `step_action`). `defer`: owed to the
        next native call -- a task epilogue then runs the whole control step as one launch (`defer_step_action`);
        the action tensor is kept alive until that call."""
        assert action.dtype == torch.float32 and action.is_contiguous() and action.shape[0] == self.num_envs
        if defer:
            self._deferred_action = action
            self._sim.defer_step_action(action.data_ptr(), action.shape[1], n_substeps, self._stream())
        else:
            self._sim.step_action(action.data_ptr(), action.shape[1], n_substeps, self._stream())

    def task_peg_outputs(self, task, obs: torch.Tensor, reward: torch.Tensor, flags: torch.Tensor, head: torch.Tensor):
        self._sim.task_peg_outputs(task, obs.data_ptr(), reward.data_ptr(), flags.data_ptr(), head.data_ptr(), self._stream())

    def task_push_outputs(self, task, obs: torch.Tensor, reward: torch.Tensor, flags: torch.Tensor):
        self._sim.task_push_outputs(task, obs.data_ptr(), reward.data_ptr(), flags.data_ptr(), self._stream())

    def task_pick_outputs(self, task, obs: torch.Tensor, reward: torch.Tensor, flags: torch.Tensor):
        self._sim.task_pick_outputs(task, obs.data_ptr(), reward.data_ptr(), flags.data_ptr(), self._stream())

    def profile_enable(self, on: bool = True):
        """bracket the solve / narrowphase launches with HIP events on their launch stream"""
        self._sim.profile_enable(on)

    def profile_read(self):
        return self._sim.profile_read()

    def overflow_count(self) -> int:
        return self._sim.overflow_count(self._stream())

    # row helpers --------------------------------------------------------------
    def body_rows(self, body_row: int) -> slice:
        """rows of `cuda_rigid_body_data` holding body `body_row` for envs 0..N-1"""
        return slice(body_row * self.num_envs, (body_row + 1) * self.num_envs)

    # sub-scene bookkeeping of the reference (sapien_env.py:1080-1096) -----------------------------------
    def set_scene_offset(self, scene, offset):
        """The reference isolates its envs by spacing them out in ONE PhysX scene. Here every env has its own
        frame, bodies of different envs can never meet, so the offset is recorded for `get_scene_offset` and
        otherwise unused."""
        self.__dict__.setdefault("_scene_offsets", {})[id(scene)] = np.asarray(offset, dtype=np.float32).reshape(3)

    def get_scene_offset(self, scene) -> np.ndarray:
        return self.__dict__.get("_scene_offsets", {}).get(id(scene), np.zeros(3, dtype=np.float32))

    # introspection: the builder-time component objects, filled in by ManiSkillScene._setup; below the env
    # layer (no builders) the lists are made from the compiled model's body names
    def _component_lists(self):
        comps = getattr(self, "_components", None)
        if comps is None and self.model is not None:
            from . import PhysxArticulationLinkComponent, PhysxRigidDynamicComponent, PhysxRigidStaticComponent

            comps = dict(dynamic=[], static=[], links=[])
            for i, name in enumerate(self.model.link_names):
                c = PhysxArticulationLinkComponent()
                c.name, c.index, c.gpu_pose_index = name, i, i * self.num_envs
                comps["links"].append(c)
            for k, name in enumerate(list(self.model.free_names) + list(self.model.kin_names)):
                c = PhysxRigidDynamicComponent()
                c.name, c.kinematic = name, k >= self.model.n_free
                c.gpu_pose_index, c.gpu_index = (self.model.n_link + k) * self.num_envs, k * self.num_envs
                comps["dynamic"].append(c)
            for name in self.model.static_names:
                c = PhysxRigidStaticComponent()
                c.name = name
                comps["static"].append(c)
            self._components = comps
        return comps or dict(dynamic=[], static=[], links=[])

    @property
    def rigid_dynamic_components(self):
        """dynamic and kinematic bodies, in `cuda_rigid_body_data` row order"""
        return list(self._component_lists()["dynamic"])

    @property
    def rigid_static_components(self):
        return list(self._component_lists()["static"])

    @property
    def articulation_link_components(self):
        return list(self._component_lists()["links"])

    def get_contacts(self, env_index: int = 0):
        """body pairs of env `env_index` with a contact in the last substep -> [PhysxContact] (the reference's CPU-only
        `px.get_contacts()`, utils/sapien_utils.py:215-260). One aggregate point per shape pair: impulse on bodies[0]
        (the core exports per-pair impulses, include/mssim.h). Synchronises."""
        from . import PhysxContact, PhysxContactPoint

        m = self.model
        cnt = self.read_internal("contact_count", max(m.n_pair, 1))[:, env_index].cpu().numpy()
        imp = self.read_internal("pair_impulse", max(3 * m.n_pair, 1))[:, env_index].cpu().numpy().reshape(-1, 3)
        lists = self._component_lists()
        by_row = {}
        for c in lists["links"] + lists["dynamic"]:
            by_row[c.gpu_pose_index // max(self.num_envs, 1)] = c
        # (a static body has no row: it is found by the name of its shape's owner)
        statics = {c.name: c for c in lists["static"]}
        world = lists["static"][0] if lists["static"] else None
        out = []
        pair_shape, shape_row = m.arrays["pair_shape"], m.arrays["shape_row"]

        def body_of(shape):
            row = int(shape_row[shape])
            return by_row[row] if row in by_row else statics.get(m.shape_owner[shape], world)

        for p in range(m.n_pair):
            if cnt[p] <= 0:
                continue
            out.append(PhysxContact(body_of(int(pair_shape[p][0])), body_of(int(pair_shape[p][1])), [PhysxContactPoint(impulse=imp[p].copy())]))
        return out

    def sync_poses_gpu_to_cpu(self):
        """viewer hook of the reference (copies poses into the CPU entities for rendering); there are no CPU
        entities here, the buffers read through `cuda_rigid_body_data` are the state"""
        self.gpu_fetch_all()
