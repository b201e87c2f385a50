#!/bin/bash
# fused vs split timing of the same library (env var MSSIM_SOLVER)
cd "$(dirname "$0")/.."
for mode in fused split; do
  echo "== $mode"
  MSSIM_SOLVER=$mode AB_CHILD=1 MSSIM_LIB=$PWD/maniskill_amd/_native/libmssim.so python scripts/ab_run.py ${1:-PickCube-v1} ${2:-4096} ${3:-100}
done
