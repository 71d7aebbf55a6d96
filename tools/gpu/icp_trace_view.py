"""The launches of the LAST f4l_patch_loop step in a kernel trace (tools/gpu/icp_trace.sh): start, end and duration of every
icp_kernel launch relative to the step's binning kernel -- which size classes overlap, which wait."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
bins = [i for i, r in enumerate(rows) if "icp_bin_patches" in r["Kernel_Name"]]
if bins:
    seg = rows[bins[-1]:]
else:
    icp = [i for i, r in enumerate(rows) if "icp_kernel" in r["Kernel_Name"]]
    seg = rows[icp[-1]:]
t0 = int(seg[0]["Start_Timestamp"])
end = t0
for r in seg:
    if "icp_kernel" not in r["Kernel_Name"] and "icp_bin" not in r["Kernel_Name"]:
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    end = max(end, e)
    print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:8.1f})  lds {r.get('LDS_Block_Size', '?'):>7s} wg {r.get('Workgroup_Size_X', '?'):>4s} grid {r.get('Grid_Size_X', '?'):>8s}  {r['Kernel_Name'][:60]}")
print("step span", (end - t0) / 1e3, "us")
