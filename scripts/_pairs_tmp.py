import sys, numpy as np, torch
import maniskill_amd.envs
import gymnasium as gym
from tests import oracle_backend as ob
ob.register("f32", "oracle_f32_env")
env = gym.make(sys.argv[1], num_envs=2, obs_mode="state", sim_backend="oracle_f32_env")
env.reset(seed=0)
px = env.unwrapped.scene.px
m = px.model; A = m.arrays
print({k: v for k, v in m.scalars.items()})
names = list(m.link_names)
st, sk, si, sb, sf = A["shape_type"], A["shape_body_kind"], A["shape_body_index"], A["shape_bound"], A["shape_frame"]
print("shape types", st.tolist())
print("kinds", sk.tolist()); print("index", si.tolist())
px.gpu_fetch_all() if hasattr(px, "gpu_fetch_all") else None
link_pose = px.cuda_rigid_body_data.torch()[:, :7].cpu().numpy() if hasattr(px.cuda_rigid_body_data, "torch") else None
print("pairs", len(A["pair_shape"]))
from collections import Counter
print(Counter((int(st[a]), int(st[b])) for a, b in A["pair_shape"]))
print("bound", sb[:, 3].round(3).tolist())
