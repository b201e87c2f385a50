"""host-side cost of ManiSkillVectorEnv.step (metrics, auto-reset) on top of env.step: cProfile over K steps
with random actions.   usage: vector_profile.py [env_id] [N]"""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import maniskill_amd.envs  # noqa
import gymnasium as gym
from maniskill_amd.vector.wrappers.gymnasium import ManiSkillVectorEnv

env_id = sys.argv[1] if len(sys.argv) > 1 else "PickCube-v1"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
K = 300
env = ManiSkillVectorEnv(gym.make(env_id, num_envs=N, obs_mode="state", control_mode="pd_joint_delta_pos"), N, ignore_terminations=False, record_metrics=True)
env.reset(seed=0)
acts = [2 * torch.rand(N, 8, device="cuda") - 1 for _ in range(K)]
for i in range(60):
    env.step(acts[i])
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(K):
    env.step(acts[i])
torch.cuda.synchronize()
print(f"{env_id} N={N}: vector env step {1e3 * (time.perf_counter() - t) / K:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for i in range(K):
    env.step(acts[i])
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40)
print(s.getvalue())
