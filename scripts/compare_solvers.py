"""long-run behaviour of the two solve kernels on the bench workload: contact statistics + timing"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import maniskill_amd.envs
import gymnasium as gym
N = 4096
env = gym.make("PickCube-v1", num_envs=N)
base = env.unwrapped
env.reset(seed=[2022 + i for i in range(N)])
px = base.scene.px
torch.manual_seed(0)
for blk in range(5):
    px.profile_enable(True)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(50):
        env.step(2 * torch.rand(N, 8, device="cuda") - 1)
    torch.cuda.synchronize(); dt = time.time() - t
    prof = px.profile_read()
    cnt = px.read_internal("contact_count", base.scene.model.n_pair)
    per_env = cnt.sum(0)
    active_pairs = (cnt > 0).float().sum(0)
    q = base.agent.robot.qpos
    print(f"steps {50*(blk+1):4d}: {dt/50*1e3:.3f} ms/step solve {prof['solve'][0]/prof['solve'][1]*1e3:.1f} us narrow {prof['narrow'][0]/prof['narrow'][1]*1e3:.1f} us | contacts/env mean {per_env.mean().item():.2f} max {per_env.max().item():.0f} pairs>0 mean {active_pairs.mean().item():.2f} | overflow {px.overflow_count()} | cube z min {base.cube.pose.p[:,2].min().item():.3f} | nan {torch.isnan(q).any().item()}")
