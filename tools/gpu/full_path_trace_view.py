"""The kernels of the LAST pipeline.full_path call of a rocprofv3 kernel trace, in launch order, runs of the same kernel merged.
Usage: full_path_trace_view.py <kernel_trace.csv> [min_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
# the last call starts at the last bbox_kernel that is followed by a knn_lanes_kernel<true>
starts = [i for i, r in enumerate(rows) if "knn_lanes_kernel<true>" in r["Kernel_Name"]]
i0 = starts[-1]
while i0 > 0 and "bbox_kernel" not in rows[i0]["Kernel_Name"]: i0 -= 1
sel = rows[i0:]
t0 = int(sel[0]["Start_Timestamp"])
out, tot = [], 0.0
for r in sel:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    n = name(r)[:90]
    if out and out[-1][0] == n: out[-1][1] += d; out[-1][2] += 1
    else: out.append([n, d, 1, (int(r["Start_Timestamp"]) - t0) / 1e3])
for n, d, c, at in out:
    if d >= min_us: print("%10.1f us at %10.1f  x%-3d %s" % (d, at, c, n))
print("kernel time %.1f us, span %.1f us" % (tot, (int(sel[-1]["End_Timestamp"]) - t0) / 1e3))
