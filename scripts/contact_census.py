"""Contact census on the CPU oracle (test infrastructure, not the product): which BODY pairs produce the contact
points of a task under random actions, and how many points an env carries at once.

    python scripts/contact_census.py [env_id] [N] [K] [control_mode]

Prints, over K control steps of N envs: the distribution of the per-env contact total (before any capacity cut:
`contact_count` is the narrowphase's own per-pair count), and for the envs above 32 points the share of every body
pair. Used to size the contact tables / design the manifold reduction."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import maniskill_amd.envs  # noqa
import gymnasium as gym
from tests import oracle_backend as ob

env_id = sys.argv[1] if len(sys.argv) > 1 else "PegInsertionSide-v1"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
K = int(sys.argv[3]) if len(sys.argv) > 3 else 200
mode = sys.argv[4] if len(sys.argv) > 4 else "pd_joint_delta_pos"

ob.register("f32", "cpu_oracle_f32")
torch.manual_seed(0)
env = gym.make(env_id, num_envs=N, obs_mode="state", control_mode=mode, sim_backend="cpu_oracle_f32")
base = env.unwrapped
env.reset(seed=0)
px = base.scene.px
model = base.scene.model
A = model.arrays
n_pair = model.n_pair
row_names = list(model.link_names) + list(model.free_names) + list(model.kin_names)


def body_of(s):
    r = int(A["shape_row"][s])
    return row_names[r] if r >= 0 else "world"


pair_body = [(body_of(int(A["pair_shape"][p][0])), body_of(int(A["pair_shape"][p][1]))) for p in range(n_pair)]
adim = base.single_action_space.shape[0]
tot_hist = []
share = {}
shape_share = {}
for t in range(K):
    env.step(2 * torch.rand(N, adim) - 1)
    cnt = px.read_internal("contact_count", n_pair).reshape(n_pair, N).numpy()
    tot = cnt.sum(0)
    tot_hist.append(tot.copy())
    for e in np.nonzero(tot > 32)[0]:
        for p in np.nonzero(cnt[:, e])[0]:
            share[pair_body[p]] = share.get(pair_body[p], 0) + int(cnt[p, e])
            key = (pair_body[p], int(A["shape_type"][int(A["pair_shape"][p][0])]), int(A["shape_type"][int(A["pair_shape"][p][1])]))
            shape_share[key] = shape_share.get(key, 0) + int(cnt[p, e])
    if (t + 1) % 100 == 0:
        env.reset()
H = np.concatenate(tot_hist)
print(f"{env_id} {mode} N={N} K={K}: contact points per env-step: mean {H.mean():.1f} p50 {np.percentile(H, 50):.0f} p90 {np.percentile(H, 90):.0f} "
      f"p99 {np.percentile(H, 99):.0f} max {H.max():.0f}; share of env-steps > 32: {(H > 32).mean():.4f}, > 48: {(H > 48).mean():.4f}, > 64: {(H > 64).mean():.4f}, > 96: {(H > 96).mean():.4f}")
print("body pairs in env-steps with more than 32 points (points summed):")
for k, v in sorted(share.items(), key=lambda kv: -kv[1])[:25]:
    print(f"  {k[0]:>24s} <-> {k[1]:<24s} {v}")
print("by shape types (0 plane 1 box 5 convex):")
for k, v in sorted(shape_share.items(), key=lambda kv: -kv[1])[:25]:
    print(f"  {k[0][0]:>24s} <-> {k[0][1]:<24s} types {k[1]},{k[2]}  {v}")
print("oracle overflow envs", px.overflow_count())
