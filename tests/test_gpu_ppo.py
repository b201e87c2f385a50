"""The PPO caller of the env API (north star: "the PPO baseline runs"): examples/ppo.py -- own code that uses the env exactly
where the reference's baseline does (examples/baselines/ppo/ppo.py:195-213: gym.make(..., num_envs, sim_backend) wrapped in
ManiSkillVectorEnv(ignore_terminations, record_metrics); :325-331: `final_info` / `_final_info` / `final_observation` after the
auto-reset) -- trained on PushCube-v1 for a fixed small budget on the HIP back end. Floor from profiles/round2/ppo_pushcube.log
(same script, same hyper-parameters: success_once 0.62 after 5.0 M steps, 0.73 after 5.8 M), with room for seed-to-seed spread.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ppo_learns_pushcube_within_a_small_budget():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "ppo.py"), "--env-id", "PushCube-v1", "--num-envs", "2048", "--total-timesteps", "6000000"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(rows) >= 50 and rows[-1]["step"] >= 5_900_000
    early = max(x.get("success_once", 0.0) for x in rows[:3])
    late = max(x.get("success_once", 0.0) for x in rows[-5:])
    print(f"PPO PushCube-v1: success_once {early:.3f} in the first iterations -> {late:.3f} after {rows[-1]['step']} steps in {rows[-1]['wall_s']} s "
          f"({rows[-1]['rollout_sps']} rollout steps/s), return {rows[-1].get('return')}")
    assert early < 0.1 and late >= 0.45, (early, late)
    assert rows[-1]["rollout_sps"] > 200_000
