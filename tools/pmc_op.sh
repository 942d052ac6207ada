#!/bin/bash
# two rocprofv3 --pmc passes over one tools/bench_ops.py group, summed per kernel: tools/pmc_op.sh <group> <kernel substring>
set -o pipefail
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pmc_op
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $out/p1 -- python3 $root/tools/bench_ops.py $1 > $out/p1.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM --output-format csv -d $out/p2 -- python3 $root/tools/bench_ops.py $1 > $out/p2.log 2>&1 || exit 1
python3 - "$2" <<'PY'
import csv, glob, collections, os, sys
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_op"
for d in ("p1", "p2"):
    f = glob.glob(out + "/" + d + "/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if sys.argv[1] not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(d, k[:70], {a: "%.4g" % b for a, b in v.items()})
PY
