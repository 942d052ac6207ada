"""Timeline of one training step from a rocprofv3 --kernel-trace CSV: per queue busy time, gaps on the main queue, what runs at the tail.
    python tools/trace_timeline.py <kernel_trace.csv> [step_index]"""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
rows.sort(key=lambda r: r["s"])
ad = [r for r in rows if r["n"].startswith("adamw")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(ad) - 3
t0, t1 = ad[k]["e"], ad[k + 1]["e"]
step = [r for r in rows if r["s"] >= t0 and r["e"] <= t1]
print("step %d: %.3f ms, %d kernels" % (k, (t1 - t0) / 1e6, len(step)))
mainq = ad[k]["Queue_Id"]
byq = collections.defaultdict(list)
for r in step: byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    busy = sum(r["e"] - r["s"] for r in rs)
    print("queue %s%s: %d kernels, busy %.3f ms, first start +%.3f, last end +%.3f" % (q, " (main)" if q == mainq else "", len(rs), busy / 1e6, (rs[0]["s"] - t0) / 1e6, (max(r["e"] for r in rs) - t0) / 1e6))
m = byq[mainq]
gaps = []
for a, b in zip(m, m[1:]):
    g = b["s"] - a["e"]
    if g > 3000: gaps.append((g, a["n"], b["n"], (a["e"] - t0) / 1e6))
print("main-queue gaps > 3 us: %d, total %.3f ms" % (len(gaps), sum(g[0] for g in gaps) / 1e6))
for g in sorted(gaps, reverse=True)[:12]: print("   %.1f us at +%.3f ms after %s before %s" % (g[0] / 1e3, g[3], g[1], g[2]))
# main-queue kernels: time while a side kernel overlaps vs not
side = [r for q, rs in byq.items() if q != mainq for r in rs]
def overlap(r):
    return sum(max(0, min(r["e"], s["e"]) - max(r["s"], s["s"])) for s in side)
ov = sum(overlap(r) for r in m)
print("main busy %.3f ms of which %.3f ms beside a side-queue kernel" % (sum(r["e"] - r["s"] for r in m) / 1e6, ov / 1e6))
# side kernels: portion not overlapped by any main kernel
def ovm(r):
    return sum(max(0, min(r["e"], s["e"]) - max(r["s"], s["s"])) for s in m)
for r in side[-6:]:
    print("   side tail: %-44s +%.3f .. +%.3f ms (%.1f us, %.1f us beside main kernels)" % (r["n"], (r["s"] - t0) / 1e6, (r["e"] - t0) / 1e6, (r["e"] - r["s"]) / 1e3, ovm(r) / 1e3))
for r in m[-8:]:
    print("   main tail: %-44s +%.3f .. +%.3f ms" % (r["n"], (r["s"] - t0) / 1e6, (r["e"] - t0) / 1e6))
# per-kernel-name durations on the main queue by whether a side kernel ran beside it (> 50 % of its time)
d = collections.defaultdict(lambda: [[], []])
for r in m:
    d[r["n"]][1 if overlap(r) > 0.5 * (r["e"] - r["s"]) else 0].append((r["e"] - r["s"]) / 1e3)
print("%-46s %8s %8s" % ("main-queue kernel: mean us", "alone", "beside"))
for n, (a, b) in sorted(d.items(), key=lambda kv: -sum(kv[1][0]) - sum(kv[1][1]))[:16]:
    print("%-46s %5.1f x%-3d %5.1f x%-3d" % (n, sum(a) / max(len(a), 1), len(a), sum(b) / max(len(b), 1), len(b)))
