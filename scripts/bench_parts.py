"""per-part kernel timing (HIP events on the launch stream): robot only / cube only / full scene"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maniskill_amd.model.compile import SceneModelBuilder
from maniskill_amd.model.scenes import panda_tabletop_model, panda_record, table_record, ground_record, cube_record
from maniskill_amd.physx.system import MssimSystem

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096

def run(name, model, q=None, steps=60):
    px = MssimSystem("cuda:0"); px.gpu_init(model, N)
    if q is not None and model.n_dof:
        px.cuda_articulation_qpos.torch()[:] = q.cuda()
        px.cuda_articulation_target_qpos.torch()[:] = q.cuda()
        px.gpu_apply_all()
    px.step(20); torch.cuda.synchronize()
    px.profile_enable(True)
    px.step(steps)
    prof = px.profile_read()
    cnt = px.read_internal("contact_count", max(model.n_pair, 1)).sum(0) if model.n_pair else torch.zeros(1)
    print(f"{name:28s} solve {prof['solve'][0]/max(prof['solve'][1],1)*1e3:8.1f} us  narrow {prof['narrow'][0]/max(prof['narrow'][1],1)*1e3:8.1f} us  contacts/env mean {cnt.float().mean().item():.2f} max {cnt.max().item():.0f}  pairs {model.n_pair}")
    px.close()

rest_up = torch.tensor([0, -np.pi / 8, 0, -np.pi * 5 / 8, 0, np.pi * 3 / 4, np.pi / 4, 0.04, 0.04])
rest_dn = torch.tensor([0, np.pi / 8, 0, -np.pi * 5 / 8, 0, np.pi * 3 / 4, np.pi / 4, 0.04, 0.04])
rec = panda_record(); rec.link_shapes = {}
b = SceneModelBuilder(); b.set_articulation(rec)
run("robot only (no shapes)", b.compile(), rest_up)
for iters in (15, 7, 1):
    b = SceneModelBuilder(); b.add_actor(table_record()); b.add_actor(ground_record()); b.add_actor(cube_record())
    run(f"cube on table, {iters} pos iters", b.compile(position_iterations=iters))
run("full scene, arm up", panda_tabletop_model(), rest_up)
run("full scene, arm near table", panda_tabletop_model(), rest_dn)
q = rest_dn.clone(); q[1] = 0.62  # fingers pressed onto the table
run("full scene, fingers on table", panda_tabletop_model(), q)
