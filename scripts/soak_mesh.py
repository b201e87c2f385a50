"""random-action soak of the mesh kernel variant (k_solve16<9, 0, true>): the Panda tabletop scene with a ribbed triangle-mesh
mat under the cube, N envs, K control steps of random joint targets (pd-style: targets move by up to 0.1 rad per step), a
state reset every 200 steps. Checks finiteness, reports overflow envs and the step rate.   usage: soak_mesh.py [N] [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from maniskill_amd.model import geom
from maniskill_amd.model.compile import ActorRecord, SceneModelBuilder, ShapeRecord
from maniskill_amd.model.scenes import cube_record, ground_record, panda_record, table_record
from maniskill_amd.physx.system import MssimSystem

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
n = 12
xs, ys = np.meshgrid(np.linspace(-0.3, 0.3, n + 1), np.linspace(-0.3, 0.3, n + 1), indexing="ij")
V = np.stack([xs.ravel(), ys.ravel(), 0.004 * np.cos(25 * xs.ravel())], 1)
F = []
for i in range(n):
    for j in range(n):
        a = i * (n + 1) + j
        F += [[a, a + n + 1, a + 1], [a + 1, a + n + 1, a + n + 2]]
b = SceneModelBuilder()
b.set_articulation(panda_record())
b.add_actor(table_record())
b.add_actor(ground_record())
b.add_actor(ActorRecord("mat", "static", [ShapeRecord("trimesh", geom.pose(), vertices=V, triangles=np.asarray(F))], initial_pose=geom.pose([0.05, 0, 0.006])))
b.add_actor(cube_record(p=(0.05, 0.0, 0.035)))
model = b.compile()
px = MssimSystem(device="cuda:0")
px.gpu_init(model, N)
rest = torch.tensor([0, np.pi / 8, 0, -np.pi * 5 / 8, 0, np.pi * 3 / 4, np.pi / 4, 0.04, 0.04], device="cuda")
row = model.row_of("cube")


def reset():
    px.cuda_articulation_qpos.torch()[:] = rest + 0.02 * torch.randn(N, 9, device="cuda")
    px.cuda_articulation_qvel.torch()[:] = 0
    px.cuda_articulation_target_qpos.torch()[:] = px.cuda_articulation_qpos.torch()
    c = px.cuda_rigid_body_data.torch()[row * N : (row + 1) * N]
    c[:] = 0
    c[:, 0] = 0.05 + 0.2 * (torch.rand(N, device="cuda") - 0.5)
    c[:, 1] = 0.2 * (torch.rand(N, device="cuda") - 0.5)
    c[:, 2] = 0.035
    c[:, 3] = 1
    px.gpu_apply_all()
    px.wake_all()


torch.manual_seed(0)
reset()
bad = 0
px.profile_enable(True)
px.profile_read()
t0 = time.perf_counter()
for k in range(1, K + 1):
    tq = px.cuda_articulation_target_qpos.torch()
    tq += 0.1 * (2 * torch.rand(N, 9, device="cuda") - 1)
    tq[:, 7:] = tq[:, 7:].clamp(-0.01, 0.04)
    px.gpu_apply_articulation_target_position()
    px.step(5)
    if k % 100 == 0:
        px.gpu_fetch_all()
        s = torch.cat([px.cuda_articulation_qpos.torch().flatten(), px.cuda_rigid_body_data.torch().flatten()])
        bad += int((~torch.isfinite(s)).sum())
        ov = px.read_internal("overflow", 1)[0].to(torch.int32)
        reasons = {b: int(((ov & b) != 0).sum()) for b in (1, 2, 4, 8, 16)}
        print("   overflow reasons (bit: envs)", reasons)
        cz = px.cuda_rigid_body_data.torch()[row * N : (row + 1) * N, 2]
        prof = px.profile_read()
        print(f"steps {k - 99:5d}-{k:5d}: {100 * N / (time.perf_counter() - t0) / 1e6:.2f} M env-steps/s, control-step kernel {prof['solve'][0] / max(prof['solve'][1], 1):.3f} ms, non-finite {bad}, overflow envs {px.overflow_count()}, "
              f"cube z min {float(cz.min()):.3f} max {float(cz.max()):.3f}, cubes below the table {int((cz < -0.05).sum())}")
        t0 = time.perf_counter()
    if k % 200 == 0:
        reset()
assert bad == 0
if os.environ.get("MSSIM_LIB", "").endswith("_clk.so"):  # (phase-clock build, scripts/phase_clocks.py --build: cycles per phase over the run)
    import ctypes
    dbg = ctypes.CDLL(os.environ["MSSIM_LIB"])
    buf = (ctypes.c_ulonglong * 32)()
    torch.cuda.synchronize()
    dbg.mssim_debug_phase_clocks(buf, 1)
    tot = sum(buf[i] for i in range(32))
    print("phase clocks (slot: share of the summed wave cycles):", {i: round(100.0 * buf[i] / tot, 1) for i in range(32) if buf[i] > 0.003 * tot})
print("soak ok")
