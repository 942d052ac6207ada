#!/bin/bash
# HBM-side traffic per kernel (separate --pmc passes, as MI355X_MICROARCH.md's HBM section prescribes):
#   tools/pmc_traffic.sh <out.json>     -- runs bench.py (3 steps) under rocprofv3 twice: FETCH_SIZE, WRITE_SIZE
out=${1:-gpurun_out/pmc_traffic.json}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
d=$root/gpurun_out/pmc_traffic
rm -rf $d && mkdir -p $d
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d $d/$c -- python3 $root/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $d/$c.log 2>&1 || { echo "pass $c failed"; tail -5 $d/$c.log; exit 1; }
done
python3 - "$d" "$root/$out" <<'PY'
import csv, glob, json, sys, collections
d, out = sys.argv[1], sys.argv[2]
workload = None      # the bench line of the profiled run names the workload these counters belong to (bench.py only echoes them for it)
for line in open("%s/FETCH_SIZE.log" % d):
    if line.startswith("{") and '"workload"' in line:
        workload = json.loads(line)["config"]["workload"]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.Counter())
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("%s/%s/*/*counter_collection.csv" % (d, c)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k][r["Counter_Name"]] += 1
res = {}
for k, v in agg.items():
    l = max(n[k].values())
    # rocprofv3 reports both in KiB; gfx950 correction: FETCH_SIZE tallies 128-byte requests as 64 bytes -> double it
    res[k] = {"launches": l, "fetch_bytes_per_launch": 2.0 * 1024 * v.get("FETCH_SIZE", 0.0) / max(n[k]["FETCH_SIZE"], 1),
              "write_bytes_per_launch": 1024 * v.get("WRITE_SIZE", 0.0) / max(n[k]["WRITE_SIZE"], 1)}
json.dump({"workload": workload, "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --steps 3 --warmup 1; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM section); bytes per launch, averaged over all launches of the kernel", "kernels": res}, open(out, "w"), indent=1, sort_keys=True)
rows = sorted(res.items(), key=lambda kv: -(kv[1]["fetch_bytes_per_launch"] + kv[1]["write_bytes_per_launch"]) * kv[1]["launches"])
named = ("warp_fuse", "wf_pack_bits", "patchify", "ce_argmax", "mask_sort", "build_rowmaps", "gather_rows")   # the gather / warp / fusion path: always listed
for i, (k, v) in enumerate(rows):
    if i < 16 or any(t in k for t in named):
        print("%-50s %5d launches  fetch %8.1f MB  write %8.1f MB" % (k[:50], v["launches"], v["fetch_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6))
PY
