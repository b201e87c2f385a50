"""host-side cost of partial resets (the auto-reset path of ManiSkillVectorEnv): cProfile over K
(step + partial reset of ~5 % of the envs) iterations.   usage: reset_profile.py [env_id] [N]"""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import maniskill_amd.envs  # noqa
import gymnasium as gym

env_id = sys.argv[1] if len(sys.argv) > 1 else "PickCube-v1"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
K = 200
env = gym.make(env_id, num_envs=N, obs_mode="state", control_mode="pd_joint_delta_pos")
env.reset(seed=0)
acts = [2 * torch.rand(N, 8, device="cuda") - 1 for _ in range(K)]
idxs = [torch.nonzero(torch.rand(N, device="cuda") < 0.05).flatten() for _ in range(K)]
for i in range(10):
    env.step(acts[i]); env.reset(options=dict(env_idx=idxs[i]))
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(K):
    env.step(acts[i])
torch.cuda.synchronize()
t_step = (time.perf_counter() - t) / K
t = time.perf_counter()
for i in range(K):
    env.step(acts[i]); env.reset(options=dict(env_idx=idxs[i]))
torch.cuda.synchronize()
t_both = (time.perf_counter() - t) / K
print(f"{env_id} N={N}: step {1e3 * t_step:.3f} ms, step + partial reset {1e3 * t_both:.3f} ms -> partial reset {1e3 * (t_both - t_step):.3f} ms")
pr = cProfile.Profile()
pr.enable()
for i in range(K):
    env.reset(options=dict(env_idx=idxs[i]))
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue())
