#!/bin/bash
# measurement: two independent B=4 training processes sharing the GPU against one B=8 process (is there throughput in kernel-level concurrency?)
root=$GRAFT_REPO_ROOT
timeout -k 10 200 python3 $root/bench.py --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one proc B=8:', d['ms_per_step'], d['value'])" || exit 1
timeout -k 10 200 python3 $root/bench.py --no-cpu-baseline --batch 4 --steps 40 --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one proc B=4:', d['ms_per_step'], d['value'])" || exit 1
(timeout -k 10 200 python3 $root/bench.py --no-cpu-baseline --batch 4 --steps 150 --warmup 30 2>/dev/null > /tmp/p1.json) &
p1=$!
(timeout -k 10 200 python3 $root/bench.py --no-cpu-baseline --batch 4 --steps 150 --warmup 30 2>/dev/null > /tmp/p2.json) &
p2=$!
wait $p1 || exit 1
wait $p2 || exit 1
python3 -c "
import json
a=json.loads(open('/tmp/p1.json').read().strip().splitlines()[-1]); b=json.loads(open('/tmp/p2.json').read().strip().splitlines()[-1])
print('two procs B=4 each:', a['ms_per_step'], b['ms_per_step'], 'sum frames/s', a['value']+b['value'])"
