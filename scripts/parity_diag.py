"""GPU diagnostics of the HIP-vs-oracle comparisons (not a test): where do per-pair contact counts differ, how do the
state errors split between envs with equal and with different counts, and which capacity overflows in the end-effector
control modes.   usage: parity_diag.py [tabletop|peg|ee] ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import maniskill_amd.envs  # noqa
import gymnasium as gym
from maniskill_amd.physx.system import MssimSystem
from tests import oracle_backend as ob
from tests import test_gpu_parity as tp

what = sys.argv[1:] or ["tabletop", "peg", "ee"]


def report(a, b, model, label):
    same = (a["cnt"] == b["cnt"]).all(0)
    same_tot = a["cnt"].sum(0) == b["cnt"].sum(0)
    dq = (a["q"] - b["q"]).abs().max(1).values
    dv = (a["qd"] - b["qd"]).abs().max(1).values
    tot = b["cnt"].sum(0)
    print(f"[{label}] envs {len(dq)}: same per-pair counts {same.float().mean():.3f}, same totals {same_tot.float().mean():.3f}; "
          f"points/env mean {tot.float().mean():.1f} max {int(tot.max())}")
    for name, m in (("same counts", same), ("same totals only", same_tot & ~same), ("different totals", ~same_tot)):
        if m.any():
            print(f"    {name:18s} n={int(m.sum()):5d}  max|dq| {dq[m].max():.2e}  p99|dq| {dq[m].quantile(0.99):.2e}  max|dqd| {dv[m].max():.2e}  p99|dqd| {dv[m].quantile(0.99):.2e}")
    light = same & (tot <= 8)
    heavy = same & (tot > 8)
    for name, m in (("same & <= 8 points", light), ("same & > 8 points", heavy)):
        if m.any():
            print(f"    {name:18s} n={int(m.sum()):5d}  max|dq| {dq[m].max():.2e}  max|dqd| {dv[m].max():.2e}")


if "tabletop" in what:
    for urdf in ("v2", "v3"):
        model = tp.panda_tabletop_model() if urdf == "v2" else tp.panda_v3_tabletop_model()
        N = 1024
        gpu, cpu = tp.make_pair(model, N)
        q, qd, tq, cube = tp.random_tabletop_state(N, 2)
        for px in (gpu, cpu):
            tp.set_state(px, model, N, q, qd, tq, cube)
            px.step(1)
        report(tp.get_state(gpu, model, N), tp.get_state(cpu, model, N), model, f"tabletop one substep, panda_{urdf}")
        raw = cpu.read_internal("raw_contact_count", 1)[0]
        print(f"    raw points/env mean {raw.mean():.1f} max {int(raw.max())}; overflow gpu {gpu.overflow_count()} cpu {cpu.overflow_count()}")

if "peg" in what:
    N = 128
    env, model, st = tp._peg_model_and_states(N, 3)
    base = env.unwrapped
    cpu = base.scene.px
    gpu = MssimSystem(device="cuda:0")
    gpu.timestep = cpu.timestep
    gpu.gpu_init(model, N)
    rb = st["rb"].clone().reshape(model.n_rows, N, 13)
    tcp = rb[model.link_names.index("panda_hand_tcp"), :, :3]
    r_box = model.row_of("box_with_hole")
    rb[r_box, :, :3] = tcp
    rb[r_box, :, 2] = tcp[:, 2] - base.peg_half_sizes[:, 0].cpu() + 0.005
    for px in (gpu, cpu):
        dev = px.device
        px.cuda_rigid_body_data.torch()[:] = rb.reshape(-1, 13).to(dev)
        px.cuda_articulation_qpos.torch()[:] = st["q"].to(dev)
        px.cuda_articulation_qvel.torch()[:] = st["qd"].to(dev)
        px.cuda_articulation_target_qpos.torch()[:] = st["q"].to(dev)
        px.gpu_apply_all()
        px.step(1)
    a, b = tp.get_state(gpu, model, N), tp.get_state(cpu, model, N)
    report(a, b, model, "peg, fingers on the box with the hole")
    A = model.arrays
    rows = list(model.link_names) + list(model.free_names) + list(model.kin_names)
    bod = lambda s_: rows[int(A["shape_row"][s_])] if int(A["shape_row"][s_]) >= 0 else "world"
    keys = sorted({(bod(int(A["pair_shape"][p][0])), bod(int(A["pair_shape"][p][1]))) for p in range(model.n_pair)})
    kid = torch.tensor([keys.index((bod(int(A["pair_shape"][p][0])), bod(int(A["pair_shape"][p][1])))) for p in range(model.n_pair)])
    ca = torch.zeros(len(keys), N).index_add_(0, kid, a["cnt"])
    cb = torch.zeros(len(keys), N).index_add_(0, kid, b["cnt"])
    print(f"    same per-BODY-pair counts: {(ca == cb).all(0).float().mean():.3f}")
    e = int(torch.nonzero(~(a['cnt'] == b['cnt']).all(0))[0]) if (~(a['cnt'] == b['cnt']).all(0)).any() else 0
    print(f"    env {e}: pairs (gpu, cpu) that differ:", [(p, int(a['cnt'][p, e]), int(b['cnt'][p, e])) for p in range(model.n_pair) if a['cnt'][p, e] != b['cnt'][p, e]][:12])
    print("    overflow gpu", gpu.overflow_count(), "cpu", cpu.overflow_count())

if "ee" in what:
    for mode in ("pd_ee_delta_pos", "pd_ee_delta_pose"):
        N = 4096
        env = gym.make("PickCube-v1", num_envs=N, sim_backend="physx_cuda", control_mode=mode)
        adim = env.unwrapped.single_action_space.shape[0]
        env.reset(seed=0)
        g = torch.Generator(device="cuda").manual_seed(0)
        px = env.unwrapped.scene.px
        model = env.unwrapped.scene.model
        seen = torch.zeros(N, dtype=torch.int32, device="cuda")
        worst = torch.zeros(N, device="cuda")
        for i in range(1000):
            env.step(2 * torch.rand(N, adim, device="cuda", generator=g) - 1)
            if i % 10 == 9:
                seen |= px.read_internal("overflow", 1)[0].int()
                worst = torch.maximum(worst, px.read_internal("contact_count", model.n_pair).sum(0))
        seen |= px.read_internal("overflow", 1)[0].int()
        print(f"[{mode}] envs with an overflow in 1000 steps: {int((seen != 0).sum())} of {N}; by reason: hits {int((seen & 1 != 0).sum())} convex {int((seen & 2 != 0).sum())} raw {int((seen & 4 != 0).sum())} contacts {int((seen & 8 != 0).sum())}; "
              f"max solved contacts seen {int(worst.max())}")
        cnt = px.read_internal("contact_count", model.n_pair)
        e = int(cnt.sum(0).argmax())
        A = model.arrays
        rows = list(model.link_names) + list(model.free_names) + list(model.kin_names)
        bod = lambda s_: rows[int(A["shape_row"][s_])] if int(A["shape_row"][s_]) >= 0 else "world"
        print("    busiest env now:", [(bod(int(A["pair_shape"][p][0])), bod(int(A["pair_shape"][p][1])), int(A["shape_type"][int(A["pair_shape"][p][0])]), int(A["shape_type"][int(A["pair_shape"][p][1])]), int(cnt[p, e])) for p in torch.nonzero(cnt[:, e]).flatten().tolist()])
        env.close()
