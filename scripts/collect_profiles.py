"""Copy the artefacts of a measurement pass from gpurun_out/ (scratch, merged back from the GPU box) into
profiles/<round>/ (tracked): bench line (+ PMC traffic patched in), rocprofv3 kernel stats, PMC summaries,
config matrix, block times.   usage: python scripts/collect_profiles.py [round1]"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles", sys.argv[1] if len(sys.argv) > 1 else "round3")
sys.path.insert(0, ROOT)
from bench import kernel_source_sha
os.makedirs(dst, exist_ok=True)


def newest(pattern):
    files = glob.glob(os.path.join(src, pattern))
    return max(files, key=os.path.getmtime) if files else None


def clean_copy(name, out=None):
    p = os.path.join(src, name)
    if os.path.exists(p):
        lines = [l for l in open(p) if "amdgpu.ids" not in l and "RuntimeWarning" not in l and "c /= stddev" not in l]
        open(os.path.join(dst, out or name), "w").writelines(lines)


f = newest("stats/*kernel_stats.csv") or newest("stats/*/*kernel_stats.csv")
if f:
    shutil.copy(f, os.path.join(dst, "bench_kernel_stats.csv"))
f = newest("stats20/*kernel_stats.csv") or newest("stats20/*/*kernel_stats.csv")
if f:  # (python bench.py --steps 20 --warmup 5: what the driver's round-end run executes)
    shutil.copy(f, os.path.join(dst, "bench_kernel_stats_steps20.csv"))
if os.path.exists(os.path.join(src, "pmc_summary.json")):
    shutil.copy(os.path.join(src, "pmc_summary.json"), os.path.join(dst, "pmc_summary.json"))
for name in ("bench_matrix.log", "block_times.log", "cpu_config1.log", "bench_series.log"):
    clean_copy(name)
clean_copy("phase_pick.log", "phase_clocks_pickcube.log")

if os.path.exists(os.path.join(src, "sq_counters.json")):
    shutil.copy(os.path.join(src, "sq_counters.json"), os.path.join(dst, "sq_counters.json"))
PROTO = "PickCube-v1 envs=4096 control_freq=20 steps=1000 warmup=5"
out = json.load(open(os.path.join(dst, "sq_counters.json")))["protocols"].get(PROTO, {}).get("counters", {}) if os.path.exists(os.path.join(dst, "sq_counters.json")) else {}

# bench line with the PMC traffic of the dominant kernel
p = os.path.join(src, "bench.json.log")
if os.path.exists(p):
    d = json.loads([l for l in open(p) if l.startswith("{")][-1])
    ks = json.load(open(os.path.join(dst, "pmc_summary.json")))["protocols"][PROTO]["kernels"]
    key = [k for k in ks if k.startswith("k_solve16<")]
    if key:
        d["roofline"]["traffic"] = ks[key[0]]["hbm_bytes_per_launch"]
        d["roofline"]["traffic_raw"] = ks[key[0]]["hbm_bytes_per_launch_raw"]
    open(os.path.join(dst, "bench.json.log"), "w").write(json.dumps(d) + "\n")
    print(d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["traffic"], d["cpu_baseline"]["value"])
for k, v in out.items():
    print(k, f"{v['per_launch']:.4g}")
