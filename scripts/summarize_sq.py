"""rocprofv3 --pmc SQ / instruction-cache passes (scripts/measure_round.sh) -> sq_counters.json: per-launch means of every
counter for the control-step kernel, stamped with the kernel source hash bench.py checks.
   usage: summarize_sq.py <dir holding pmc_ic/ and pmc_sq/> <out.json>"""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_sha

root, out_path = sys.argv[1:3]
protocol = sys.argv[3] if len(sys.argv) > 3 else None  # bench.protocol_of(args) of the passes (scripts/measure_round.sh)
out = {}
for d in ("pmc_ic", "pmc_sq"):
    files = glob.glob(os.path.join(root, d, "*", "*_counter_collection.csv"))
    if not files:
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(files[0])):
        if r["Kernel_Name"].startswith("void k_solve16"):
            a = agg[r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        out[k] = dict(launches=n, per_launch=v / n)
doc = dict(kernel_source_sha=kernel_source_sha(), protocols={})
if os.path.exists(out_path):
    try:
        old = json.load(open(out_path))
        if old.get("kernel_source_sha") == doc["kernel_source_sha"]:
            doc["protocols"] = old.get("protocols", {})
    except Exception:
        pass
doc["note"] = ("rocprofv3 --pmc, control-step kernel k_solve16<9, TASK> (whole env.step), bench.py with the flags of the protocol; "
               "two passes (instruction counts + instruction cache, SQ cycles); SQ_*_CYCLES / SQ_WAIT_* are quad-cycles summed over waves")
doc["protocols"][protocol] = dict(counters=out)
json.dump(doc, open(out_path, "w"), indent=1)
for k, v in sorted(out.items()):
    print(k, f"{v['per_launch']:.4g}")
