"""The Fetch driving around procedurally generated rooms: `SceneManipulation-v1` with the `SyntheticRooms` scene builder
(one of five triangle-mesh layouts per sub-scene). Random arm / head actions, the base drives forward and turns away when
it stops making progress.   usage: python examples/fetch_rooms.py [num_envs] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import mani_skill.envs  # noqa: F401  (alias of maniskill_amd)
import gymnasium as gym

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 400
env = gym.make("SceneManipulation-v1", num_envs=N, obs_mode="state", build_config_idxs=[i % 5 for i in range(N)])
obs, _ = env.reset(seed=0)
base = env.unwrapped
builder = base.scene_builder
print("layouts:", builder.build_configs, "| scenery:", sorted(builder.scene_objects), "| action space:", base.single_action_space)
turn = torch.zeros(N, device=base.device)
last_xy = base.agent.robot.get_qpos()[:, :2].clone()
travelled = torch.zeros(N, device=base.device)
for t in range(K):
    a = 0.3 * (2 * torch.rand(N, 13, device=base.device) - 1)
    a[:, 11] = 0.8               # forward
    a[:, 12] = turn              # yaw rate
    obs, _, _, _, info = env.step(a)
    if t % 10 == 9:              # stuck against something: turn for a while
        xy = base.agent.robot.get_qpos()[:, :2]
        moved = torch.linalg.norm(xy - last_xy, dim=1)
        travelled += moved
        turn = torch.where(moved < 0.1, torch.full_like(turn, 0.6), torch.zeros_like(turn))
        last_xy = xy.clone()
reasons = base.scene.px.read_internal("overflow", 1)[0].to(torch.int32)
print("overflow reasons (bit: envs)", {b: int(((reasons & b) != 0).sum()) for b in (1, 2, 4, 8, 16)})
print(f"{K} steps of {N} envs: mean distance travelled {float(travelled.mean()):.2f} m, envs that reported a capacity overflow: {base.scene.px.overflow_count()}")
