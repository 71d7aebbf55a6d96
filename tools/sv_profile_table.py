"""Per-kernel table of one partition profile: time per call from the kernel stats CSV, HBM bytes per call from the FETCH_SIZE /
WRITE_SIZE passes (tools/gpu/sv_round.sh).  Usage: python tools/sv_profile_table.py <dir> [calls]"""
import csv, json, sys, os, collections
d = sys.argv[1]; calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t = collections.defaultdict(float); c = collections.defaultdict(int)
for r in csv.DictReader(open(os.path.join(d, "partition_10M_kernel_stats.csv"))):
    name = r["Name"].split("(")[0].replace("void ", "")[:60]
    t[name] += float(r["TotalDurationNs"]) / calls / 1e6; c[name] += int(r["Calls"]) // calls
by = {}
p = os.path.join(d, "svp_bytes.json")
if os.path.exists(p):
    by = json.load(open(p))["per_call_by_kernel"]
rows = []
for k in set(t) | set(by):
    if not (k.startswith("f4l::") or "rocprim" in k): continue
    f = by.get(k, {}).get("FETCH_SIZE", 0) * 1024 * 2 / 1e9; w = by.get(k, {}).get("WRITE_SIZE", 0) * 1024 / 1e9
    rows.append((t.get(k, 0), c.get(k, 0), f, w, k))
rows.sort(reverse=True)
print("%8s %6s %8s %8s %8s  %s" % ("ms/call", "launch", "fetch GB", "write GB", "total GB", "kernel"))
for r in rows:
    print("%8.3f %6d %8.2f %8.2f %8.2f  %s" % (r[0], r[1], r[2], r[3], r[2] + r[3], r[4]))
print("%8.3f %6d %8.2f %8.2f %8.2f  TOTAL" % (sum(r[0] for r in rows), sum(r[1] for r in rows), sum(r[2] for r in rows), sum(r[3] for r in rows), sum(r[2] + r[3] for r in rows)))
