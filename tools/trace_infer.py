"""One inference step from a rocprofv3 --kernel-trace CSV: kernels in order with duration and the gap in front of each.
    python tools/trace_infer.py <kernel_trace.csv> [first-kernel-substring]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
rows.sort(key=lambda r: r["s"])
key = sys.argv[2] if len(sys.argv) > 2 else "mask_sort"
starts = [i for i, r in enumerate(rows) if key in r["n"]]
i0, i1 = starts[-3], starts[-2]
step = rows[i0:i1]
print("step: %.1f us wall, %d kernels, busy %.1f us" % ((rows[i1]["s"] - rows[i0]["s"]) / 1e3, len(step), sum(r["e"] - r["s"] for r in step) / 1e3))
prev = None
for r in step:
    gap = (r["s"] - prev) / 1e3 if prev else 0.0
    print("%7.1f us  gap %5.1f  %s  grid %s wg %s" % ((r["e"] - r["s"]) / 1e3, gap, r["n"], r["Grid_Size_X"], r["Workgroup_Size_X"]))
    prev = r["e"]
