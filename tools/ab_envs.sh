#!/bin/bash
# A/B of environment settings inside one GPU call: tools/ab_envs.sh "A=1 B=2" "A=0" ...   (each argument: one setting, run through bench.py)
set -o pipefail
root=$GRAFT_REPO_ROOT
for setting in "$@"; do
  echo "== $setting"
  env $setting timeout -k 10 200 python3 $root/bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ms_per_step', d['ms_per_step'], 'value', d['value'], [(f['kernel'], f['ms_per_step'], f['launches_per_step']) for f in d.get('families', [])][:3])" || exit 1
done
