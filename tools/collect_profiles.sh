#!/bin/bash
# Collect the round's profile set on the GPU box (one gpurun call): tools/collect_profiles.sh <tag>   e.g. r02_a
# Writes under gpurun_out/<tag>/ ; copy what is to be judged into profiles/.
set -o pipefail
tag=${1:-r02_a}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --profile-all > $out/bench.json 2> $out/bench.err || exit 1
echo "bench done"; tail -c 300 $out/bench.json | head -c 300; echo
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/stats.log 2>&1 ) || exit 1
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/train_b8_kernel_stats.csv && echo "stats done"
bash tools/pmc_traffic.sh gpurun_out/$tag/pmc_traffic.json > $out/pmc_traffic_summary.txt 2>&1 || { tail -5 $out/pmc_traffic_summary.txt; exit 1; }
echo "pmc traffic done"
bash tools/pmc_mfma.sh gpurun_out/$tag/pmc_mfma.json > $out/pmc_mfma_summary.txt 2>&1 || { tail -5 $out/pmc_mfma_summary.txt; exit 1; }
echo "pmc mfma done"
: > $out/bench_other_configs.jsonl
for flags in "--dtype fp32" "--mode infer --batch 16" "--mode infer --batch 16 --dtype bf16x3" "--mode infer --batch 16 --dtype fp32" "--mode infer --batch 1" "--mode infer --batch 1 --dtype bf16x3" "--mode infer --batch 1 --dtype fp32" \
             "--time_stamp 3" "--time_stamp 3 --mask random --mask_ratio 0.75" "--time_stamp 3 --mask random --mask_ratio 0.9" \
             "--time_stamp 2 --mask random --mask_ratio 0.9" "--com vqstar" \
             "--model amortized_joint_patch8_shallow --batch 4 --steps 10" "--model amortized_joint_patch8_shallow --batch 4 --mode infer --steps 10"; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 $flags >> $out/bench_other_configs.jsonl 2>> $out/bench_other.err || exit 1
  echo "config done: $flags"
done
# the grouped weight-gradient and convolution kernels alone, and the grouped kernel's HBM-side traffic
timeout -k 10 200 python tools/bench_ops.py wgroup conv > $out/bench_ops_wgroup_conv.txt 2>&1 || exit 1
bash tools/pmc_wgroup.sh > $out/pmc_wgroup_summary.txt 2>&1 || { tail -5 $out/pmc_wgroup_summary.txt; exit 1; }
echo "wgroup done"
