"""Replay recorded trajectories, by actions or by environment states, on any simulation backend
(counterpart of mani_skill/trajectory/replay_trajectory.py:110-390: `--use-env-states`,
`--use-first-env-state`, `--sim-backend`). Episodes are replayed one after another in a
single-env instance of the task; the result lists, per episode, the final success flag and the
largest deviation of the replayed env states from the recorded ones.

    python -m maniskill_amd.trajectory.replay_trajectory --traj-path demos/trajectory.npz --use-first-env-state
"""
import argparse
import json
import os
from typing import List, Optional

import numpy as np
import torch

import maniskill_amd.envs  # noqa: F401
import gymnasium as gym

from maniskill_amd.trajectory import utils as trajectory_utils


def _to_torch_state(state: dict, device) -> dict:
    return {k: {n: torch.as_tensor(v, device=device)[None] for n, v in d.items()} for k, d in state.items()}


def _state_diff(a: dict, b: dict) -> float:
    worst = 0.0
    for k in a:
        for n in a[k]:
            worst = max(worst, float(np.abs(np.asarray(a[k][n]) - np.asarray(b[k][n])).max()))
    return worst


def replay(traj_path: str, sim_backend: Optional[str] = None, use_env_states: bool = False, use_first_env_state: bool = False,
           count: Optional[int] = None, env_kwargs: Optional[dict] = None) -> List[dict]:
    json_path = os.path.splitext(traj_path)[0] + ".json"
    with open(json_path) as f:
        meta = json.load(f)
    data = trajectory_utils.load_h5_data(traj_path)
    info = meta["env_info"]
    kw = dict(info.get("env_kwargs") or {})
    kw.pop("num_envs", None)
    if sim_backend is not None:
        kw["sim_backend"] = sim_backend
    kw.update(env_kwargs or {})
    results = []
    env = None
    for ep in meta["episodes"][: count if count is not None else None]:
        traj = data[f"traj_{ep['episode_id']}"]
        kw["control_mode"] = ep["control_mode"]
        if env is None or env.unwrapped.control_mode != ep["control_mode"]:
            if env is not None:
                env.close()
            env = gym.make(info["env_id"], num_envs=1, **kw)
        base = env.unwrapped
        env.reset(seed=ep["episode_seed"])
        states = trajectory_utils.dict_to_list_of_dicts(traj["env_states"]) if "env_states" in traj else None
        if use_first_env_state or use_env_states:
            assert states is not None, "the trajectory holds no env_states"
            base.set_state_dict(_to_torch_state(states[0], base.device))
            base.agent.controller.reset()
        worst = 0.0
        info_step = {}
        for t, a in enumerate(traj["actions"]):
            _, _, _, _, info_step = env.step(torch.as_tensor(a, device=base.device)[None])
            if states is not None:
                now = {k: {n: v[0].cpu().numpy() for n, v in d.items()} for k, d in base.get_state_dict().items()}
                worst = max(worst, _state_diff(now, states[t + 1]))
                if use_env_states:
                    base.set_state_dict(_to_torch_state(states[t + 1], base.device))
        success = bool(info_step["success"][0]) if "success" in info_step else None
        results.append(dict(episode_id=ep["episode_id"], elapsed_steps=len(traj["actions"]), success=success,
                            recorded_success=ep.get("success"), max_state_deviation=worst))
    if env is not None:
        env.close()
    return results


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--traj-path", required=True)
    ap.add_argument("-b", "--sim-backend", default=None)
    ap.add_argument("--use-env-states", action="store_true")
    ap.add_argument("--use-first-env-state", action="store_true")
    ap.add_argument("--count", type=int, default=None)
    a = ap.parse_args()
    for r in replay(a.traj_path, a.sim_backend, a.use_env_states, a.use_first_env_state, a.count):
        print(json.dumps(r))


if __name__ == "__main__":
    main()
