"""env-steps/s of the BASELINE.json single-GPU configurations, step-only and step+reset (reset every 200
steps, as the reference's gpu_sim.py:96-106,166-178), one line per configuration."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import maniskill_amd.envs  # noqa
import gymnasium as gym

def run(env_id, N, steps=200, control_mode="pd_joint_delta_pos", **kw):
    torch.manual_seed(2022)
    env = gym.make(env_id, num_envs=N, obs_mode="state", control_mode=control_mode, **kw)
    base = env.unwrapped
    adim = base.single_action_space.shape[0]
    env.reset(seed=2022)
    for _ in range(10):
        env.step(2 * torch.rand(N, adim, device="cuda") - 1)
    env.reset(seed=2022)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps):
        env.step(2 * torch.rand(N, adim, device="cuda") - 1)
    torch.cuda.synchronize(); dt_step = time.perf_counter() - t
    env.reset(seed=2022)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(2 * steps):
        env.step(2 * torch.rand(N, adim, device="cuda") - 1)
        if (i + 1) % 200 == 0:
            env.reset()
    torch.cuda.synchronize(); dt_reset = time.perf_counter() - t
    out = dict(env_id=env_id, num_envs=N, control_mode=control_mode, substeps=base._sim_steps_per_control, step_only=round(N * steps / dt_step), step_reset_every_200=round(N * 2 * steps / dt_reset),
               ms_per_step=round(dt_step / steps * 1e3, 3), overflow_envs=base.scene.px.overflow_count())
    print(json.dumps(out), flush=True)
    env.close()

only = sys.argv[1:]
for args, kw in ((("PickCube-v1", 4096), {}), (("PushCube-v1", 4096), {}), (("PegInsertionSide-v1", 2048), {}),
                 (("PickCube-v1", 4096), dict(control_mode="pd_ee_delta_pos")), (("PickCube-v1", 4096), dict(control_mode="pd_ee_delta_pose")),
                 (("PickCube-v1", 16384), {}),
                 (("PickCube-v1", 4096), dict(sim_config=dict(control_freq=25)))):  # 4 substeps (SURVEY 8d reports 5 and 4)
    if not only or args[0] in only or kw.get("control_mode") in only or ("substeps4" in only and "sim_config" in kw):
        run(*args, **kw)
