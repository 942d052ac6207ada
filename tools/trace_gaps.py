#!/usr/bin/env python3
"""GPU idle time per training step from a rocprofv3 --kernel-trace CSV: union of kernel intervals over all streams vs wall
time between optimizer launches.  Caveat: the tracer slows the host enough to make the step host-bound (a 250 us gap
appears behind the optimizer that event timing without the tracer does not show: 7-9 us); read the per-kernel gaps, not
the step-boundary one.  usage: trace_gaps.py <kernel_trace.csv> [steps]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:44]) for r in rows)
ad = [i for i, e in enumerate(ev) if e[2].startswith("adamw")]
seg = ev[ad[-nsteps - 1] + 1:ad[-1] + 1]
t0, t1 = seg[0][0], seg[-1][1]
busy, cs, ce, gaps = 0, seg[0][0], seg[0][1], []
for s, e, n in seg[1:]:
    if s > ce:
        busy += ce - cs
        gaps.append((s - ce, n))
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("per step: wall %.3f ms, busy %.3f ms, idle %.3f ms in %.0f gaps" % ((t1 - t0) / nsteps / 1e6, busy / nsteps / 1e6, (t1 - t0 - busy) / nsteps / 1e6, len(gaps) / nsteps))
gaps.sort(reverse=True)
print("largest gaps (us):", [(round(g / 1e3, 1), n[:28]) for g, n in gaps[:8]])
c = collections.Counter()
for g, n in gaps:
    c[n] += g
print("idle by following kernel (us/step):", [(n[:34], round(v / 1e3 / nsteps, 1)) for n, v in c.most_common(10)])
