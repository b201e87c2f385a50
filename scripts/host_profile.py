"""host-side cost of env.step: cProfile over K steps (GPU work is asynchronous, so tottime here is
Python + launch overhead, the part a faster kernel cannot hide)"""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import maniskill_amd.envs  # noqa
import gymnasium as gym

env_id = sys.argv[1] if len(sys.argv) > 1 else "PickCube-v1"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = 200
env = gym.make(env_id, num_envs=N, obs_mode="state", control_mode="pd_joint_delta_pos")
env.reset(seed=0)
acts = [2 * torch.rand(N, 8, device="cuda") - 1 for _ in range(K)]
for a in acts[:20]:
    env.step(a)
torch.cuda.synchronize()
t = time.time()
for a in acts:
    env.step(a)
t_issue = time.time() - t
torch.cuda.synchronize()
t_all = time.time() - t
print(f"{env_id} N={N}: host issue {t_issue/K*1e3:.3f} ms/step, wall {t_all/K*1e3:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for a in acts:
    env.step(a)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
print(s.getvalue())
