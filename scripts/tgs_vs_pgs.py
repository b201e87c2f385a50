"""PGS (the product's solver: 15 position + 1 velocity Gauss-Seidel iterations per substep, contact multipliers warm-started
from the previous substep) against the same solver started cold (MSSIM_REF_COLD=1) and against a TGS-style position
sub-stepping variant of the oracle (MSSIM_REF_TGS=1, oracle/oracle_sim.cpp) on the contact scenarios the tasks depend on:
cube at rest, two-cube stack, drop, grasp-and-lift with the Panda, a hull slab on the table. CPU only (the oracle);
one child process per solver because the switch is read when a system is created.   usage: tgs_vs_pgs.py [child]"""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure():
    import numpy as np
    import torch
    from maniskill_amd.model.compile import SceneModelBuilder
    from maniskill_amd.model.scenes import cube_record, ground_record, panda_tabletop_model, table_record
    from tests import oracle_backend as ob
    from tests.test_oracle_contacts import _slab_on_table

    out = {}
    # cube at rest: penetration and residual velocity after 0.3 s (awake)
    b = SceneModelBuilder(); b.add_actor(table_record()); b.add_actor(ground_record()); b.add_actor(cube_record())
    model = b.compile(sleep_threshold=0.0)
    px = ob.make_system(model, 1)
    px.step(100); px.gpu_fetch_all()
    s = px.cuda_rigid_body_data.torch()[model.row_of("cube")]
    out["rest: sink below 0.02 m [um]"] = round((0.02 - s[2].item()) * 1e6, 2)
    out["rest: |v| after 1 s [mm/s]"] = round(s[7:10].norm().item() * 1e3, 4)
    # stack
    b = SceneModelBuilder(); b.add_actor(table_record()); b.add_actor(cube_record(name="c0", p=(0, 0, 0.02))); b.add_actor(cube_record(name="c1", p=(0.005, 0.003, 0.06)))
    model = b.compile(sleep_threshold=0.0)
    px = ob.make_system(model, 1)
    zs = []
    for _ in range(200):
        px.step(1); px.gpu_fetch_all(); zs.append(px.cuda_rigid_body_data.torch()[model.row_of("c1"), 2].item())
    rb = px.cuda_rigid_body_data.torch()
    out["stack: top cube sink [um]"] = round((0.06 - zs[-1]) * 1e6, 2)
    out["stack: top cube drift xy [um]"] = round(float(((rb[model.row_of("c1"), 0] - 0.005) ** 2 + (rb[model.row_of("c1"), 1] - 0.003) ** 2).sqrt()) * 1e6, 2)
    out["stack: z jitter over the last 1 s [um]"] = round((max(zs[100:]) - min(zs[100:])) * 1e6, 3)
    # drop from 10 cm: deepest penetration and rebound
    b = SceneModelBuilder(); b.add_actor(table_record()); b.add_actor(ground_record()); b.add_actor(cube_record())
    model = b.compile(sleep_threshold=0.0)
    px = ob.make_system(model, 1)
    px.cuda_rigid_body_data.torch()[model.row_of("cube"), 2] = 0.12
    px.gpu_apply_all()
    zmin, zmax_after = 1.0, 0.0
    landed = False
    for i in range(100):
        px.step(1); px.gpu_fetch_all()
        z = px.cuda_rigid_body_data.torch()[model.row_of("cube"), 2].item()
        zmin = min(zmin, z)
        if z < 0.0205:
            landed = True
        if landed:
            zmax_after = max(zmax_after, z)
    out["drop 0.1 m: deepest penetration [um]"] = round((0.02 - zmin) * 1e6, 1)
    out["drop 0.1 m: rebound height [um]"] = round((zmax_after - 0.02) * 1e6, 1)
    # hull slab
    model = _slab_on_table(True)
    px = ob.make_system(model, 1)
    px.step(60); px.gpu_fetch_all()
    s = px.cuda_rigid_body_data.torch()[model.row_of("slab")]
    out["hull slab: tilt after 0.6 s [mdeg]"] = round(float(np.rad2deg(2 * np.arccos(min(1.0, abs(s[3].item()))))) * 1e3, 2)
    # grasp and lift (tests/test_oracle_contacts.py::test_panda_grasp_holds_cube)
    model = panda_tabletop_model()
    px = ob.make_system(model, 1)
    rest = np.array([0, np.pi / 8, 0, -np.pi * 5 / 8, 0, np.pi * 3 / 4, np.pi / 4, 0.04, 0.04], dtype=np.float32)
    tcp = model.link_names.index("panda_hand_tcp")

    def tcp_pos(qv):
        px.cuda_articulation_qpos.torch()[:] = qv
        px.gpu_apply_articulation_qpos(); px.gpu_update_articulation_kinematics(); px.gpu_fetch_articulation_link_pose()
        return px.cuda_rigid_body_data.torch()[tcp, :3].clone()

    def ik(q0, target):
        q, idx = q0.clone(), [1, 3, 5]
        for _ in range(30):
            p = tcp_pos(q)
            J = torch.zeros(3, 3)
            for k, j in enumerate(idx):
                dq = q.clone(); dq[0, j] += 1e-4
                J[:, k] = (tcp_pos(dq) - p) / 1e-4
            step = torch.linalg.solve(J.T @ J + 1e-6 * torch.eye(3), J.T @ (target - p))
            for k, j in enumerate(idx):
                q[0, j] += step[k]
        return q

    q = ik(torch.from_numpy(rest).clone()[None], torch.tensor([0.0, 0.0, 0.02]))
    q_lift = ik(q, torch.tensor([0.0, 0.0, 0.15]))
    px.cuda_articulation_qpos.torch()[:] = q; px.cuda_articulation_target_qpos.torch()[:] = q; px.gpu_apply_all()
    tq = q.clone(); tq[0, 7:] = -0.01
    px.cuda_articulation_target_qpos.torch()[:] = tq; px.gpu_apply_articulation_target_position(); px.step(50)
    px.gpu_fetch_all()
    z0 = px.cuda_rigid_body_data.torch()[model.row_of("cube"), 2].item() - tcp_pos(px.cuda_articulation_qpos.torch().clone())[2].item()
    for i in range(100):
        a = (i + 1) / 100
        tq = q * (1 - a) + q_lift * a; tq[0, 7:] = -0.01
        px.cuda_articulation_target_qpos.torch()[:] = tq; px.gpu_apply_articulation_target_position(); px.step(1)
    px.step(50); px.gpu_fetch_all()
    cube = px.cuda_rigid_body_data.torch()[model.row_of("cube")]
    z1 = cube[2].item() - px.cuda_rigid_body_data.torch()[tcp, 2].item()
    out["grasp: cube height after the lift [mm]"] = round(cube[2].item() * 1e3, 2)
    out["grasp: slip in the fingers during the lift [um]"] = round(abs(z1 - z0) * 1e6, 1)
    out["grasp: finger opening while holding [mm]"] = [round(x * 1e3, 3) for x in px.cuda_articulation_qpos.torch()[0, 7:].tolist()]
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        measure()
    else:
        res = {}
        variants = (("PGS cold", dict(MSSIM_REF_TGS="0", MSSIM_REF_COLD="1")),
                    ("PGS warm (product)", dict(MSSIM_REF_TGS="0", MSSIM_REF_COLD="0")),
                    ("TGS-style cold", dict(MSSIM_REF_TGS="1", MSSIM_REF_COLD="1")))
        for name, flags in variants:
            r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, **flags), capture_output=True, text=True, check=True)
            res[name] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        names = [v[0] for v in variants]
        print(f"{'':52s} " + " ".join(f"{n:>22s}" for n in names))
        for k in res[names[0]].keys():
            print(f"{k:52s} " + " ".join(f"{str(res[n][k]):>22s}" for n in names))
