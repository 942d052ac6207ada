#!/bin/bash
# usage: tools/pmc.sh <tag> <run_one args...>   -- separate rocprofv3 --pmc passes, results under gpurun_out/pmc_<tag>/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES" \
           "GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/tools/run_one.py "$@" > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob("$out/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:50]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    if "at::" in k or "rocclr" in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s %.4g" % (c, v))
PY
