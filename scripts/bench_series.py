"""How the control-step kernel's time develops over an unreset random-action run (the reference protocol times 1000
steps without a reset, gpu_sim.py:96-106): mean kernel time (HIP events) and contact load per window of W steps.
   usage: bench_series.py [env_id] [N] [K] [W] [control_mode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import maniskill_amd.envs  # noqa
import gymnasium as gym

env_id = sys.argv[1] if len(sys.argv) > 1 else "PickCube-v1"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
W = int(sys.argv[4]) if len(sys.argv) > 4 else 100
mode = sys.argv[5] if len(sys.argv) > 5 else "pd_joint_delta_pos"
env = gym.make(env_id, num_envs=N, control_mode=mode)
base = env.unwrapped
adim = base.single_action_space.shape[0]
env.reset(seed=[2022 + i for i in range(N)])
torch.manual_seed(2022)
px = base.scene.px
model = base.scene.model
A = model.arrays
types = torch.tensor([[int(A["shape_type"][int(A["pair_shape"][p][0])]), int(A["shape_type"][int(A["pair_shape"][p][1])])] for p in range(model.n_pair)], device="cuda")
is_mpr = ~((types[:, 0] == 0) | ((types[:, 0] == 1) & (types[:, 1] == 1)))
for w in range(K // W):
    px.profile_enable(True)
    for _ in range(W):
        env.step(2 * torch.rand(N, adim, device="cuda") - 1)
    ms, n = px.profile_read()["solve"]
    px.profile_enable(False)
    cnt = px.read_internal("contact_count", model.n_pair)
    tot = cnt.sum(0)
    mpr = (cnt[is_mpr] > 0).sum(0)
    per_block = tot.reshape(-1, 4).max(1).values
    print(f"steps {w * W + 1:5d}-{(w + 1) * W:5d}: kernel {ms / n:.4f} ms | contacts/env mean {tot.mean():.2f} p99 {tot.quantile(0.99):.0f} max {int(tot.max())} | "
          f"hull pairs in contact/env mean {mpr.float().mean():.2f} max {int(mpr.max())} | envs with > 8 contacts {int((tot > 8).sum())} | overflow {px.overflow_count()}", flush=True)
