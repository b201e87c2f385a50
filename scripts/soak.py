"""long random-action soak: K control steps of N envs with a full reset every 200 steps (and a partial reset of ~2 %
of the envs every 10 steps); checks finiteness / bounds of obs and reward throughout and prints the step rate of every
1000-step block (drift in the rate = growing contact load).   usage: soak.py [env_id] [N] [K] [control_mode] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import maniskill_amd.envs  # noqa
import gymnasium as gym

env_id = sys.argv[1] if len(sys.argv) > 1 else "PickCube-v1"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
mode = sys.argv[4] if len(sys.argv) > 4 else "pd_joint_delta_pos"
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 0
torch.manual_seed(seed)
extra = {"robot_uids": os.environ["MS_ROBOT"]} if os.environ.get("MS_ROBOT") else {}  # (MS_ROBOT=fetch with Empty-v1)
if os.environ.get("MS_SCENE_BUILDER"):  # (MS_SCENE_BUILDER=SyntheticRoomsCrowded with SceneManipulation-v1)
    extra["scene_builder_cls"] = os.environ["MS_SCENE_BUILDER"]
env = gym.make(env_id, num_envs=N, obs_mode="state", control_mode=mode, **extra)
base = env.unwrapped
adim = base.single_action_space.shape[0]
env.reset(seed=seed)
bad = torch.zeros((), dtype=torch.int64, device="cuda")
omax = torch.zeros((), device="cuda")
t0 = time.perf_counter()
for i in range(1, K + 1):
    obs, rew, term, trunc, info = env.step(2 * torch.rand(N, adim, device="cuda") - 1)
    bad += (~torch.isfinite(obs)).sum() + (~torch.isfinite(rew)).sum()
    omax = torch.maximum(omax, obs.abs().max())
    if i % 200 == 0:
        env.reset()
    elif i % 10 == 0:
        env.reset(options=dict(env_idx=torch.nonzero(torch.rand(N, device="cuda") < 0.02).flatten()))
    if i % 1000 == 0:
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{env_id} {mode} N={N} steps {i - 999}-{i}: {1000 * N / dt / 1e6:.2f} M env-steps/s, non-finite values so far {int(bad)}, max |obs| {float(omax):.2f}, overflow envs {base.scene.px.overflow_count()}", flush=True)
        t0 = time.perf_counter()
assert int(bad) == 0 and float(omax) < 1e3
print("soak ok")
