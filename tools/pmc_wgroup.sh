#!/bin/bash
# HBM-side traffic of the grouped weight-gradient launches alone (tools/bench_ops.py wgroup): two rocprofv3 --pmc passes
root=$GRAFT_REPO_ROOT
d=$root/gpurun_out/pmc_wgroup
rm -rf $d; mkdir -p $d
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $d/$c -- python3 $root/tools/bench_ops.py wgroup > $d/$c.log 2>&1 || { tail -3 $d/$c.log; exit 1; }
done
python3 - $d <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("%s/%s/*/*counter_collection.csv" % (d, c)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "gemm_tn" in k or "wgrad" in k:
                acc[(k, r["Grid_Size"])][c].append(float(r["Counter_Value"]))
for (k, g), v in sorted(acc.items()):
    f = 2 * 1024 * sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1)
    w = 1024 * sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1)
    print("%-46s grid %8s  %3d launches  fetch %7.1f MB  write %6.1f MB" % (k[:46], g, len(v["FETCH_SIZE"]), f / 1e6, w / 1e6))
PY
