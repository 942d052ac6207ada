#!/bin/bash
# A/B of an environment switch inside one GPU call: tools/ab_env.sh VAR "<bench_ops groups>" v1 v2 ...
set -o pipefail
root=$GRAFT_REPO_ROOT
var=$1; groups=$2; shift 2
for v in "$@"; do
  export $var=$v
  echo "== $var=$v"
  if [ -n "$groups" ]; then timeout -k 10 120 python3 $root/tools/bench_ops.py $groups 2>&1 | grep -v "^$\|amdgpu.ids" || exit 1; fi
  timeout -k 10 200 python3 $root/bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ms_per_step', d['ms_per_step'], 'value', d['value'])" || exit 1
done
