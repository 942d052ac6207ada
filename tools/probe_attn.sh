#!/bin/bash
# measurement only: attention backward with parts compiled out (tools/probe_attn<k>.so), per-kernel times + PMC of the real build
set -o pipefail
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/probe
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for k in 0 1 2 3; do
  if [ $k = 0 ]; then unset STAR_HIP_LIB; else export STAR_HIP_LIB=$root/tools/probe_attn$k.so; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p$k -- python3 $root/tools/bench_ops.py attn > $out/p$k.log 2>&1 || exit 1
  f=$(ls $out/p$k/*/*kernel_stats.csv | head -1)
  echo "== probe $k"; grep -E "attn" $f | cut -d, -f1-4 | cut -c1-150
done
unset STAR_HIP_LIB
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $out/pmc1 -- python3 $root/tools/bench_ops.py attn > $out/pmc1.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INST_CYCLES_VMEM --output-format csv -d $out/pmc2 -- python3 $root/tools/bench_ops.py attn > $out/pmc2.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/probe"
for d in ("pmc1", "pmc2"):
    f = glob.glob(out + "/" + d + "/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "attn" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    for k, v in acc.items():
        print(d, k[:60], {a: "%.3g" % b for a, b in v.items()})
PY
