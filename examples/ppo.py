"""Minimal PPO on the vectorised env (state observations), everything on one GPU.

Own implementation in the style the reference's baseline uses the env API
(examples/baselines/ppo/ppo.py:195-213, 321-331 of the reference): `gym.make(env_id,
num_envs=N, sim_backend="physx_cuda")`, `ManiSkillVectorEnv(env, ignore_terminations=...,
record_metrics=True)`, `final_info` / `_final_info` / `final_observation` after auto-resets.
No tyro / tensorboard dependency: plain argparse, prints a JSON line per iteration.

    python examples/ppo.py --env-id PushCube-v1 --num-envs 2048 --total-timesteps 3000000
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch
import torch.nn as nn

import mani_skill.envs  # noqa: F401  (alias of maniskill_amd; registers the tasks)
import gymnasium as gym
from mani_skill.vector.wrappers.gymnasium import ManiSkillVectorEnv


def layer_init(layer, std=np.sqrt(2), bias_const=0.0):
    torch.nn.init.orthogonal_(layer.weight, std)
    torch.nn.init.constant_(layer.bias, bias_const)
    return layer


class Agent(nn.Module):
    def __init__(self, obs_dim, act_dim):
        super().__init__()
        self.critic = nn.Sequential(
            layer_init(nn.Linear(obs_dim, 256)), nn.Tanh(), layer_init(nn.Linear(256, 256)), nn.Tanh(),
            layer_init(nn.Linear(256, 256)), nn.Tanh(), layer_init(nn.Linear(256, 1)),
        )
        self.actor_mean = nn.Sequential(
            layer_init(nn.Linear(obs_dim, 256)), nn.Tanh(), layer_init(nn.Linear(256, 256)), nn.Tanh(),
            layer_init(nn.Linear(256, 256)), nn.Tanh(), layer_init(nn.Linear(256, act_dim), std=0.01 * np.sqrt(2)),
        )
        self.actor_logstd = nn.Parameter(torch.ones(1, act_dim) * -0.5)

    def get_value(self, x):
        return self.critic(x)

    def get_action_and_value(self, x, action=None):
        mean = self.actor_mean(x)
        std = torch.exp(self.actor_logstd.expand_as(mean))
        dist = torch.distributions.Normal(mean, std)
        if action is None:
            action = dist.sample()
        return action, dist.log_prob(action).sum(1), dist.entropy().sum(1), self.critic(x)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env-id", default="PushCube-v1")
    ap.add_argument("--num-envs", type=int, default=2048)
    ap.add_argument("--num-steps", type=int, default=50)
    ap.add_argument("--total-timesteps", type=int, default=3_000_000)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--gamma", type=float, default=0.8)
    ap.add_argument("--gae-lambda", type=float, default=0.9)
    ap.add_argument("--update-epochs", type=int, default=4)
    ap.add_argument("--num-minibatches", type=int, default=32)
    ap.add_argument("--clip-coef", type=float, default=0.2)
    ap.add_argument("--ent-coef", type=float, default=0.0)
    ap.add_argument("--vf-coef", type=float, default=0.5)
    ap.add_argument("--max-grad-norm", type=float, default=0.5)
    ap.add_argument("--target-kl", type=float, default=0.1)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--partial-reset", action="store_true", help="stop episodes on success (default: ignore terminations)")
    args = ap.parse_args()

    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    device = torch.device("cuda")
    env = gym.make(args.env_id, num_envs=args.num_envs, obs_mode="state", control_mode="pd_joint_delta_pos", sim_backend="physx_cuda")
    envs = ManiSkillVectorEnv(env, args.num_envs, ignore_terminations=not args.partial_reset, record_metrics=True)
    obs_dim = envs.single_observation_space.shape[0]
    act_dim = envs.single_action_space.shape[0]
    agent = Agent(obs_dim, act_dim).to(device)
    opt = torch.optim.Adam(agent.parameters(), lr=args.lr, eps=1e-5)

    N, T = args.num_envs, args.num_steps
    obs_buf = torch.zeros((T, N, obs_dim), device=device)
    act_buf = torch.zeros((T, N, act_dim), device=device)
    logp_buf = torch.zeros((T, N), device=device)
    rew_buf = torch.zeros((T, N), device=device)
    done_buf = torch.zeros((T, N), device=device)
    val_buf = torch.zeros((T, N), device=device)
    batch = N * T
    mb = batch // args.num_minibatches
    iters = args.total_timesteps // batch
    next_obs, _ = envs.reset(seed=args.seed)
    next_done = torch.zeros(N, device=device)
    global_step = 0
    t_start = time.time()
    sim_time = 0.0
    for it in range(1, iters + 1):
        final_values = torch.zeros((T, N), device=device)
        succ, rets = [], []
        t0 = time.time()
        for t in range(T):
            global_step += N
            obs_buf[t], done_buf[t] = next_obs, next_done
            with torch.no_grad():
                a, lp, _, v = agent.get_action_and_value(next_obs)
            act_buf[t], logp_buf[t], val_buf[t] = a, lp, v.flatten()
            next_obs, r, term, trunc, info = envs.step(torch.clamp(a, -1, 1))
            next_done = torch.logical_or(term, trunc).float()
            rew_buf[t] = r
            if "final_info" in info:
                m = info["_final_info"]
                ep = info["final_info"]["episode"]
                succ.append(ep["success_once"][m].float())
                rets.append(ep["return"][m])
                with torch.no_grad():
                    final_values[t, m] = agent.get_value(info["final_observation"][m]).flatten()
        sim_time += time.time() - t0
        with torch.no_grad():
            next_value = agent.get_value(next_obs).flatten()
            adv = torch.zeros_like(rew_buf)
            last = 0
            for t in reversed(range(T)):
                nv = next_value if t == T - 1 else val_buf[t + 1]
                nnd = 1.0 - (next_done if t == T - 1 else done_buf[t + 1])
                # bootstrap through time-limit resets with the value of the final observation
                real_next = nnd * nv + final_values[t]
                delta = rew_buf[t] + args.gamma * real_next - val_buf[t]
                adv[t] = last = delta + args.gamma * args.gae_lambda * nnd * last
            ret = adv + val_buf
        b_obs, b_act = obs_buf.reshape(-1, obs_dim), act_buf.reshape(-1, act_dim)
        b_logp, b_adv, b_ret, b_val = logp_buf.reshape(-1), adv.reshape(-1), ret.reshape(-1), val_buf.reshape(-1)
        stop = False
        for _ in range(args.update_epochs):
            perm = torch.randperm(batch, device=device)
            for s in range(0, batch, mb):
                idx = perm[s : s + mb]
                _, nlp, ent, nv = agent.get_action_and_value(b_obs[idx], b_act[idx])
                logratio = nlp - b_logp[idx]
                ratio = logratio.exp()
                with torch.no_grad():
                    kl = ((ratio - 1) - logratio).mean()
                if kl > args.target_kl:
                    stop = True
                    break
                a_ = b_adv[idx]
                a_ = (a_ - a_.mean()) / (a_.std() + 1e-8)
                pg = torch.max(-a_ * ratio, -a_ * torch.clamp(ratio, 1 - args.clip_coef, 1 + args.clip_coef)).mean()
                vl = 0.5 * ((nv.flatten() - b_ret[idx]) ** 2).mean()
                loss = pg - args.ent_coef * ent.mean() + args.vf_coef * vl
                opt.zero_grad()
                loss.backward()
                nn.utils.clip_grad_norm_(agent.parameters(), args.max_grad_norm)
                opt.step()
            if stop:
                break
        out = dict(iter=it, step=global_step, wall_s=round(time.time() - t_start, 1), rollout_sps=round(global_step / max(sim_time, 1e-9)))
        if succ:
            out["success_once"] = round(torch.cat(succ).mean().item(), 3)
            out["return"] = round(torch.cat(rets).mean().item(), 3)
        print(json.dumps(out), flush=True)
    envs.close()


if __name__ == "__main__":
    main()
