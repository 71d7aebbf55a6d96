"""Timeline of the last segmentation in a rocprofv3 kernel trace (tools/gpu/sv_trace.sh): kernels above a duration, totals per name."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "init_state_kernel" in r["Kernel_Name"]]
seg = rows[idx[-1]:]
t0 = int(seg[0]["Start_Timestamp"])
tot = collections.OrderedDict()
for r in seg:
    name = r["Kernel_Name"].split("(")[0].replace("f4l::svg::", "").replace("void ", "")[:44]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d = (e - s) / 1e3
    c = tot.setdefault(name, [0, 0.0]); c[0] += 1; c[1] += d
    if d > thr:
        print(f"{(s - t0) / 1e3:9.1f} us  {name:44s} {d:8.1f} us")
print("span", (int(seg[-1]["End_Timestamp"]) - t0) / 1e3, "us; kernels", len(seg))
for k, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"   {k:44s} x{c:4d} {d:9.1f} us")
