#!/bin/bash
# the soak over every configuration (run through gpurun from the repo root; the log goes to gpurun_out/soak.log)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/soak.log
: > $O
run() { echo "== $*" >> $O; timeout -k 10 900 python3 scripts/soak.py "$@" >> $O 2>&1 || echo "FAILED: $*" >> $O; }
run PickCube-v1 4096 5000 pd_joint_delta_pos
run PickCube-v1 4096 3000 pd_ee_delta_pos
run PickCube-v1 4096 3000 pd_ee_delta_pose
run PushCube-v1 4096 3000 pd_joint_delta_pos
run PegInsertionSide-v1 2048 3000 pd_joint_delta_pos
MS_ROBOT=fetch run Empty-v1 1024 2000 pd_joint_delta_pos
run SceneManipulation-v1 1024 3000 pd_joint_delta_pos
MS_SCENE_BUILDER=SyntheticRoomsCrowded run SceneManipulation-v1 1024 2000 pd_joint_delta_pos
grep -c "soak ok" $O; grep -n "FAILED\|Error\|assert" $O | head
