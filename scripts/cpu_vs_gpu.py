"""is env.step host-bound or device-bound? enqueue time vs completion time, + cProfile of the host side"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import maniskill_amd.envs
import gymnasium as gym
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = gym.make("PickCube-v1", num_envs=N)
env.reset(seed=0)
acts = [2 * torch.rand(N, 8, device="cuda") - 1 for _ in range(100)]
for a in acts[:10]: env.step(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for a in acts: env.step(a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"N={N}: host enqueue {1e3*(t1-t0)/100:.3f} ms/step, total {1e3*(t2-t0)/100:.3f} ms/step")
pr = cProfile.Profile(); pr.enable()
for a in acts: env.step(a)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
