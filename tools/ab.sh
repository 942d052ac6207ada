#!/bin/bash
# A/B of kernel builds inside one GPU call: tools/ab.sh "<bench_ops groups>" lib1.so lib2.so ...   ("-" = the in-tree build)
set -o pipefail
root=$GRAFT_REPO_ROOT
groups=$1; shift
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset STAR_HIP_LIB; else export STAR_HIP_LIB=$root/$lib; fi
  echo "== $lib"
  timeout -k 10 120 python3 $root/tools/bench_ops.py $groups 2>&1 | grep -v "^$" || exit 1
  timeout -k 10 200 python3 $root/bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ms_per_step', d['ms_per_step'], 'value', d['value'])" || exit 1
done
