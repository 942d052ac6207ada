#!/bin/bash
# MFMA utilisation per kernel family inside the bench step (one rocprofv3 --pmc pass, no trace domains):
#   tools/pmc_mfma.sh <out.json>
# util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs); cross-check: SQ_INSTS_MFMA x 16 cycles
# (v_mfma_f32_16x16x32_bf16, MI355X_MICROARCH.md cycle table) over the same denominator.
out=${1:-gpurun_out/pmc_mfma.json}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
d=$root/gpurun_out/pmc_mfma
rm -rf $d && mkdir -p $d
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $d/p -- python3 $root/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $d/p.log 2>&1 || { echo "pass failed"; tail -5 $d/p.log; exit 1; }
python3 - "$d" "$root/$out" <<'PY'
import csv, glob, json, sys, collections
d, out = sys.argv[1], sys.argv[2]
workload = None      # the bench line of the profiled run names the workload these counters belong to (bench.py only echoes them for it)
for line in open("%s/p.log" % d):
    if line.startswith("{") and '"workload"' in line:
        workload = json.loads(line)["config"]["workload"]
fam = [("gemm_nt", ("void gemm_nt_",)), ("gemm_wgrad", ("void gemm_tn_",)), ("gemm_res_ln", ("void gemm_res_ln_",)), ("attn_fwd", ("void attn_fwd_",)),
       ("attn_bwd", ("void attn_bwd_",)), ("vq_assign", ("void vq_assign_",))]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob("%s/p/*/*counter_collection.csv" % d):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for name, pre in fam:
            if k.startswith(pre):
                agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    n[name] += 1
res = {}
for name, v in agg.items():
    simd_cycles = v["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4
    res[name] = {"launches": n[name], "SQ_VALU_MFMA_BUSY_CYCLES": v["SQ_VALU_MFMA_BUSY_CYCLES"], "SQ_INSTS_MFMA": v["SQ_INSTS_MFMA"],
                 "SQ_BUSY_CYCLES": v["SQ_BUSY_CYCLES"], "GRBM_GUI_ACTIVE": v["GRBM_GUI_ACTIVE"],
                 "mfma_util_busy_cycles": v["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles if simd_cycles else None,
                 "mfma_util_insts_x16": v["SQ_INSTS_MFMA"] * 16.0 / simd_cycles if simd_cycles else None}
json.dump({"workload": workload, "note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE over bench.py --steps 3 --warmup 1; "
                   "sums over all launches of a family; denominator = GRBM_GUI_ACTIVE / 8 x 1024 SIMDs (the counter is summed over the 8 XCDs)",
           "families": res}, open(out, "w"), indent=1, sort_keys=True)
for name, v in res.items():
    print("%-12s launches %4d  MFMA busy %5.1f %%   (insts x 16 cyc: %5.1f %%)" % (name, v["launches"], 100 * (v["mfma_util_busy_cycles"] or 0), 100 * (v["mfma_util_insts_x16"] or 0)))
PY
