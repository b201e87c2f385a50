"""raw physics timing below the env layer: apply targets -> n substeps -> fetch"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maniskill_amd.model.scenes import panda_tabletop_model
from maniskill_amd.physx.system import MssimSystem

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sub = int(sys.argv[3]) if len(sys.argv) > 3 else 5
model = panda_tabletop_model()
px = MssimSystem("cuda:0"); px.gpu_init(model, N)
rest = torch.tensor([0, np.pi / 8, 0, -np.pi * 5 / 8, 0, np.pi * 3 / 4, np.pi / 4, 0.04, 0.04], device="cuda")
px.cuda_articulation_qpos.torch()[:] = rest + 0.02 * torch.randn(N, 9, device="cuda")
r = model.row_of("cube")
cube = px.cuda_rigid_body_data.torch()[r * N:(r + 1) * N]
cube[:, :2] = 0.2 * torch.rand(N, 2, device="cuda") - 0.1
px.gpu_apply_all()
def one():
    a = 2 * torch.rand(N, 8, device="cuda") - 1
    tq = px.cuda_articulation_qpos.torch().clone()
    tq[:, :7] += 0.1 * a[:, :7]
    tq[:, 7:] = (a[:, 7:8] * 0.025 + 0.015)
    px.cuda_articulation_target_qpos.torch()[:] = tq
    px.gpu_apply_articulation_target_position()
    px.step(sub)
    px.gpu_fetch_all()
for _ in range(10): one()
torch.cuda.synchronize(); t = time.time()
for _ in range(steps): one()
torch.cuda.synchronize(); dt = time.time() - t
print(f"N={N} substeps={sub}: {dt/steps*1e3:.3f} ms/step, {N*steps/dt:,.0f} env-steps/s (physics only), overflow={px.overflow_count()}")
cnt = px.read_internal("contact_count", model.n_pair).sum(0)
print("contacts/env mean", cnt.mean().item(), "max", cnt.max().item())
