"""step rate of Empty-v1 with the Fetch (15 velocity components, generic-topology kernel, no native action map: the
ego-centric base controller needs the yaw) -- BASELINE config 5's robot and env count on an empty scene.
usage: bench_fetch.py [N] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import maniskill_amd.envs  # noqa
import gymnasium as gym

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 500
env = gym.make("Empty-v1", robot_uids="fetch", num_envs=N, obs_mode="state")
env.reset(seed=0)
acts = [2 * torch.rand(N, 13, device="cuda") - 1 for _ in range(16)]
for i in range(20):
    env.step(acts[i % 16])
px = env.unwrapped.scene.px
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    env.step(acts[i % 16])
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
px.profile_enable(True)
px.profile_read()
for i in range(50):
    env.step(acts[i % 16])
torch.cuda.synchronize()
prof = px.profile_read()
px.profile_enable(False)
kernel_ms = prof["solve"][0] / max(prof["solve"][1], 1)
print(f"Empty-v1 fetch N={N}: {N * K / dt / 1e6:.3f} M env-steps/s ({1e3 * dt / K:.3f} ms per env.step), host issue {1e3 * t_issue / K:.3f} ms, control-step kernel {kernel_ms:.3f} ms, overflow envs {px.overflow_count()}, "
      f"finite {bool(torch.isfinite(env.unwrapped.agent.robot.get_qpos()).all())}")
