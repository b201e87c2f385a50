"""Soak-run debugger: like soak.py, but keeps the last states of every env and, at the first non-finite or huge
observation, prints the offending env's recent history (flat env state rows, joint targets, actions).
usage: find_blowup.py env_id N K control_mode seed"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import maniskill_amd.envs  # noqa
import gymnasium as gym

env_id, N, K, mode, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
torch.manual_seed(seed)
env = gym.make(env_id, num_envs=N, obs_mode="state", control_mode=mode)
base = env.unwrapped
px = base.scene.px
adim = base.single_action_space.shape[0]
env.reset(seed=seed)
hist = []
fmt = lambda t: ["%.3g" % float(x) for x in t]
for i in range(1, K + 1):
    a = 2 * torch.rand(N, adim, device="cuda") - 1
    obs, rew, term, trunc, info = env.step(a)
    hist.append((base.get_state().clone(), px.cuda_articulation_target_qpos.torch().clone(), a))
    hist = hist[-10:]
    big = (~torch.isfinite(obs)) | (obs.abs() > 1e3)
    if big.any():
        e = int(big.any(1).nonzero().flatten()[0])
        print("first non-finite / huge observation at step", i, "env", e, "obs columns", big[e].nonzero().flatten().tolist()[:40], "overflow envs", px.overflow_count())
        nd = base.agent.robot.max_dof
        for k, (s_, t_, a_) in enumerate(hist):
            r = s_[e]
            print("t-%d" % (len(hist) - 1 - k), "actors", fmt(r[: len(r) - 13 - 2 * nd]))
            print("     qpos", fmt(r[-2 * nd : -nd]), "qvel", fmt(r[-nd:]))
            print("     target", fmt(t_[e]), "action", fmt(a_[e]))
        break
    if i % 200 == 0:
        env.reset()
    elif i % 10 == 0:
        env.reset(options=dict(env_idx=torch.nonzero(torch.rand(N, device="cuda") < 0.02).flatten()))
else:
    print("nothing non-finite or huge in", K, "steps")
