#!/bin/bash
# tools/diag_env.sh <mask> "ENV=.." "ENV=.." ...: bench_ops wgroup of one diagnostic build (0 = the in-tree build) under several settings
root=$(cd "$(dirname "$0")/.." && pwd)
m=$1; shift
if [ $m != 0 ]; then export STAR_HIP_LIB=$root/star_amd/diag/libstar_hip_d$m.so; fi
for setting in "$@"; do
  echo "== mask $m $setting"
  env $setting timeout -k 10 120 python3 $root/tools/bench_ops.py wgroup 2>&1 | grep grouped || exit 1
done
