"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes)
-> profiles/<round>/pmc_summary.json with per-launch means for the simulation kernels.

    python scripts/summarize_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/round2/pmc_summary.json
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB. Caveat from the guide: on gfx950 FETCH_SIZE
reads exactly 1/2 of the bytes of a wide (16 B/lane) coalesced stream and is uncalibrated for other
access widths; these kernels issue 4-byte strided accesses, so the raw value is kept and the
possible 2x under-count of the read side is stated next to it.
"""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_sha

fetch_dir, write_dir, out = sys.argv[1:4]
protocol = sys.argv[4] if len(sys.argv) > 4 else None  # bench.protocol_of(args) of the passes (scripts/measure_round.sh)
res = collections.defaultdict(dict)
for d, key in ((fetch_dir, "FETCH_SIZE"), (write_dir, "WRITE_SIZE")):
    f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != key:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        if k.startswith("k_"):
            res[k][key + "_KB_per_launch"] = v / n
            res[k]["launches_" + key] = n
for k, v in res.items():
    v["hbm_bytes_per_launch_raw"] = 1024.0 * (v.get("FETCH_SIZE_KB_per_launch", 0.0) + v.get("WRITE_SIZE_KB_per_launch", 0.0))
    # MI355X_MICROARCH.md (HBM): gfx950 tallies a 128-byte read request as 64 bytes -- double FETCH_SIZE before comparing with a byte count
    v["hbm_bytes_per_launch"] = 1024.0 * (2.0 * v.get("FETCH_SIZE_KB_per_launch", 0.0) + v.get("WRITE_SIZE_KB_per_launch", 0.0))
# one file holds the passes of several protocols (the default run: 1000 unreset steps; the driver's: 20 steps over fresh episodes):
# {"kernel_source_sha", "protocols": {protocol: {"kernels": ...}}}; an entry of other kernel sources is dropped
doc = dict(kernel_source_sha=kernel_source_sha(), protocols={})
if os.path.exists(out):
    try:
        old = json.load(open(out))
        if old.get("kernel_source_sha") == doc["kernel_source_sha"]:
            doc["protocols"] = old.get("protocols", {})
    except Exception:
        pass
doc["note"] = "rocprofv3 counters, KB per launch; hbm_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction, MI355X_MICROARCH.md, HBM), _raw = their plain sum"
doc["protocols"][protocol] = dict(kernels=res)
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
