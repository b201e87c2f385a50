"""step rate of SceneManipulation-v1 (Fetch in the SyntheticRooms layouts, one of five triangle-mesh rooms per sub-scene)
at BASELINE config 5's env count.   usage: bench_scene.py [N] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import maniskill_amd.envs  # noqa
import gymnasium as gym

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 400
env = gym.make("SceneManipulation-v1", num_envs=N, obs_mode="state", build_config_idxs=[i % 5 for i in range(N)])
env.reset(seed=0)
px = env.unwrapped.scene.px
acts = [2 * torch.rand(N, 13, device="cuda") - 1 for _ in range(16)]
for i in range(20):
    env.step(acts[i % 16])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    env.step(acts[i % 16])
    if (i + 1) % 200 == 0:
        env.reset()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
px.profile_enable(True)
px.profile_read()
for i in range(50):
    env.step(acts[i % 16])
torch.cuda.synchronize()
prof = px.profile_read()
print(f"SceneManipulation-v1 (fetch, SyntheticRooms) N={N}: {N * K / dt / 1e6:.3f} M env-steps/s ({1e3 * dt / K:.3f} ms per env.step, reset every 200; host issue {1e3 * t_issue / K:.3f} ms), "
      f"control-step kernel {prof['solve'][0] / max(prof['solve'][1], 1):.3f} ms, overflow envs {px.overflow_count()}, finite {bool(torch.isfinite(env.unwrapped.agent.robot.get_qpos()).all())}")
