"""time k_solve16 / k_narrow (HIP events) for every maniskill_amd/_native/libmssim_exp_*.so variant,
one subprocess per variant (MSSIM_LIB selects the library). usage: ab_run.py [env_id] [N] [steps]"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("AB_CHILD"):
    sys.path.insert(0, ROOT)
    import torch
    import maniskill_amd.envs  # noqa
    import gymnasium as gym
    env_id, N, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    env = gym.make(env_id, num_envs=N, obs_mode="state", control_mode="pd_joint_delta_pos")
    env.reset(seed=0)
    px = env.unwrapped.scene.px
    for _ in range(10):
        env.step(2 * torch.rand(N, 8, device="cuda") - 1)
    env.reset(seed=0)
    px.profile_enable(True)
    for _ in range(steps):
        env.step(2 * torch.rand(N, 8, device="cuda") - 1)
    torch.cuda.synchronize()
    print(os.path.basename(os.environ["MSSIM_LIB"]), px.profile_read(), flush=True)
    sys.exit(0)
args = sys.argv[1:] + ["PickCube-v1", "4096", "100"][len(sys.argv) - 1:]
for lib in sorted(glob.glob(os.path.join(ROOT, "maniskill_amd", "_native", "libmssim_exp_*.so"))):
    env = dict(os.environ, AB_CHILD="1", MSSIM_LIB=lib)
    subprocess.run([sys.executable, os.path.abspath(__file__)] + args, env=env, check=False)
