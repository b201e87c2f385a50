"""Which capacity an env exceeds, and when: `steps` random-action control steps without a reset, the MSSIM_OVERFLOW_* bits
(include/mssim.h) of every env read every 50 steps.   usage: overflow_census.py [env_id] [N] [steps] [key=value ...]"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import maniskill_amd.envs  # noqa
import gymnasium as gym

env_id = sys.argv[1] if len(sys.argv) > 1 else "SceneManipulation-v1"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
K = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
kw = dict(a.split("=", 1) for a in sys.argv[4:])
if env_id == "SceneManipulation-v1":
    kw["build_config_idxs"] = [i % 5 for i in range(N)]
torch.manual_seed(2022)
env = gym.make(env_id, num_envs=N, obs_mode="state", **kw)
base = env.unwrapped
px = base.scene.px
adim = base.single_action_space.shape[0]
env.reset(seed=2022)
names = {1: "HITS", 2: "CONVEX", 4: "RAW", 8: "CONTACTS", 16: "TRI"}
total = collections.Counter()
envs_over = set()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(K):
    env.step(2 * torch.rand(N, adim, device="cuda") - 1)
    if (i + 1) % 50 == 0:
        bits = px.read_internal("overflow", 1)[0].cpu().to(torch.int64)
        px.overflow_count()
        nz = torch.nonzero(bits).flatten().tolist()
        if nz:
            c = collections.Counter()
            for e in nz:
                envs_over.add(e)
                for b, nm in names.items():
                    if int(bits[e]) & b:
                        c[nm] += 1
                        total[nm] += 1
            print(f"steps {i - 48}..{i + 1}: {len(nz)} envs over a capacity {dict(c)} e.g. {[(e, e % 5, int(bits[e])) for e in nz[:6]]}", flush=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{env_id} {kw.get('scene_builder_cls', '')} N={N} {K} steps: {N * K / dt / 1e6:.3f} M env-steps/s; envs that ever exceeded a capacity: {len(envs_over)}; by capacity (env x 50-step windows): {dict(total)}; finite {bool(torch.isfinite(base.agent.robot.get_qpos()).all())}")
