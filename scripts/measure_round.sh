#!/bin/bash
# One measurement pass on the GPU box (run through gpurun from the repo root); everything lands in gpurun_out/ and is
# copied into profiles/<round>/ by scripts/collect_profiles.py afterwards.  usage: bash scripts/measure_round.sh [quick]
# rocprofv3: the program itself after `--` (python3), counters in passes of their own (MI355X_MICROARCH.md, rocprofv3 PMC).
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=$PWD/gpurun_out
P=/tmp/mssim_prof   # raw rocprofv3 output stays on the box (tens of MB per pass); only the summaries travel back
mkdir -p "$O"
rm -rf "$P" "$O"/stats "$O"/stats20 "$O"/pmc_fetch "$O"/pmc_write "$O"/pmc_sq "$O"/pmc_ic; mkdir -p "$P"
echo "== bench (default flags: 1000 steps)"; python3 bench.py > "$O"/bench.json.log 2> "$O"/bench.err || exit 1
tail -c 600 "$O"/bench.json.log
echo "== kernel stats"; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$P"/stats -- python3 "$OLDPWD"/bench.py --no-cpu-baseline > "$O"/bench_prof.log 2>&1) || exit 1
# (and of the command the driver's round-end run uses: 20 steps over fresh episodes)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$P"/stats20 -- python3 "$OLDPWD"/bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$O"/bench_prof20.log 2>&1) || exit 1
echo "== PMC passes"
# (the counter passes run the SAME command as a timed run, so that a launch counted is a launch timed: once with the default flags --
# 1000 unreset steps -- and once with the flags the driver's round-end run uses -- 20 steps over fresh episodes; bench.py reports the
# counters of whichever protocol it is run under)
rm -f "$O"/pmc_summary.json "$O"/sq_counters.json
for FL in "" "--steps 20 --warmup 5"; do
  if [ -z "$FL" ]; then PROTO="PickCube-v1 envs=4096 control_freq=20 steps=1000 warmup=5"; else PROTO="PickCube-v1 envs=4096 control_freq=20 steps=20 warmup=5"; fi
  rm -rf "$P"/pmc_fetch "$P"/pmc_write "$P"/pmc_ic "$P"/pmc_sq
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$P"/pmc_fetch -- python3 "$OLDPWD"/bench.py $FL --no-cpu-baseline > "$O"/pmc_fetch.log 2>&1) || exit 1
  (cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$P"/pmc_write -- python3 "$OLDPWD"/bench.py $FL --no-cpu-baseline > "$O"/pmc_write.log 2>&1) || exit 1
  (cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d "$P"/pmc_ic -- python3 "$OLDPWD"/bench.py $FL --no-cpu-baseline > "$O"/pmc_ic.log 2>&1) || exit 1
  (cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d "$P"/pmc_sq -- python3 "$OLDPWD"/bench.py $FL --no-cpu-baseline > "$O"/pmc_sq.log 2>&1) || exit 1
  python3 scripts/summarize_pmc.py "$P"/pmc_fetch "$P"/pmc_write "$O"/pmc_summary.json "$PROTO" > /dev/null || exit 1
  python3 scripts/summarize_sq.py "$P" "$O"/sq_counters.json "$PROTO" > /dev/null || exit 1
done
mkdir -p "$O"/stats "$O"/stats20 && cp "$P"/stats/*/*kernel_stats.csv "$O"/stats/ && cp "$P"/stats20/*/*kernel_stats.csv "$O"/stats20/ || exit 1
if [ "$1" != "quick" ]; then
  echo "== config matrix"; python3 scripts/bench_matrix.py > "$O"/bench_matrix.log 2>&1; grep "^{" "$O"/bench_matrix.log | cut -c1-240
  echo "== kernel time over an unreset run"; python3 scripts/bench_series.py PickCube-v1 4096 1000 100 > "$O"/bench_series.log 2>&1; tail -3 "$O"/bench_series.log
fi
echo done
