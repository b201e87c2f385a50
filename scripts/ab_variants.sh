#!/bin/bash
# build experiment variants of the HIP library (timing A/B only; results of EXP_* builds are wrong by design)
set -e
cd "$(dirname "$0")/.."
# usage: ab_variants.sh name:flags ...   e.g.  ab_variants.sh base: it1:-DEXP_ITERS=1
for v in "base:-DMSSIM_ONLY_PANDA" "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -Wno-unused-value -Wno-pass-failed -fno-hip-fp32-correctly-rounded-divide-sqrt $flags \
    -o maniskill_amd/_native/libmssim_exp_$name.so maniskill_amd/csrc/mssim_kernels.hip
done
