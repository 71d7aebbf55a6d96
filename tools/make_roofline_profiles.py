#!/usr/bin/env python3
"""Turn the raw counter summaries of tools/gpu/pmc_passes.py (run on the MI355X box, merged back under gpurun_out/) into the
files bench.py reads for `roofline*.traffic` and `roofline*.valu`:

    profiles/icp_kernel_counters.json   <- gpurun_out/<dir>/icp.json    (bench.py --config C4_50M_100k, kernel `icp_kernel`)
    profiles/knn_counters.json          <- gpurun_out/<dir>/knn.json    (tools/gpu/knn_only.py 10000000, every f4l:: / rocprim kernel of f4l_knn)
    profiles/supervoxel_counters.json   <- gpurun_out/<dir>/svp.json    (tools/gpu/svp_only.py 10000000 3)
    profiles/supervoxel_exact_counters.json <- gpurun_out/<dir>/svx.json (tools/gpu/svx_only.py 10000000 3: the default partition)

HBM bytes: FETCH_SIZE (KB) x 1024 x 2 (the gfx950 correction of MI355X_MICROARCH.md: the counter tallies 128-byte requests at
64 B) + WRITE_SIZE (KB) x 1024.  Vector issue: SQ_INSTS_VALU wave-instructions, of which the float64 ones (ADD / MUL / FMA /
TRANS _F64) cost 4 SIMD cycles and the others 2 (SIMD-32, wave64).  Each file carries the hash of the kernel sources it was
taken on; bench.py ignores it once they change.  Usage: make_roofline_profiles.py gpurun_out/<dir> <tag>"""
import hashlib
import json
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (KERNEL_SOURCES and the hash)


def summarise(c, per_call=False):
    g = (c.get("per_call") or c) if per_call else c
    f64 = sum(g.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
    insts = g.get("SQ_INSTS_VALU", 0.0)
    out = dict(fetch_bytes=int(g.get("FETCH_SIZE", 0.0) * 1024 * 2), write_bytes=int(g.get("WRITE_SIZE", 0.0) * 1024))
    out["hbm_bytes_per_launch"] = out["fetch_bytes"] + out["write_bytes"]
    if insts:
        out["valu"] = dict(insts_per_launch=int(insts), f64_insts_per_launch=int(f64), f64_share=round(f64 / insts, 4),
                           issue_cycles_per_launch=int((insts - f64) * 2 + f64 * 4),
                           weights="SIMD cycles per wave64 instruction: 4 for float64 (ADD/MUL/FMA/TRANS_F64), 2 for every other")
    out["raw"] = {k: v for k, v in g.items() if k.startswith(("SQ_", "FETCH", "WRITE", "ns"))}
    return out


def main():
    src, tag = sys.argv[1], sys.argv[2]
    jobs = [("icp.json", "icp_kernel", "icp_kernel_counters.json", "icp", "C4_50M_100k", None, 2,
             "icp_kernel: the two launches of one f4l_patch_loop step (bulk class + border class, side by side); figures per STEP"),
            ("knn.json", "ALL", "knn_counters.json", "knn", "f4l_knn k=30", 10_000_000, None,
             "every kernel of one f4l_knn call on 10 M points (binning sorts, cell tables, knn_lanes_kernel, knn_listed_kernel)"),
            ("svp.json", "ALL", "supervoxel_counters.json", "supervoxel", "f4l_supervoxel_parallel k=30", 10_000_000, None,
             "every kernel of one f4l_supervoxel_parallel call on 10 M points (kNN + normals + the ~270 launches of the segmentation)"),
            ("svx.json", "ALL", "supervoxel_exact_counters.json", "supervoxel_exact", "f4l_supervoxel k=30", 10_000_000, None,
             "every kernel of one f4l_supervoxel call on 10 M points (kNN + normals + the passes of the reference's segmentation: svx::)")]
    for raw_name, key, out_name, which, workload, units, group, what in jobs:
        path = os.path.join(src, raw_name)
        if not os.path.exists(path):
            print("missing", path)
            continue
        raw = json.load(open(path))
        if key == "ALL":  # the sum over the kernel families of one call
            tot = {}
            for fam in raw["counters"].values():
                for c, v in (fam.get("per_call") or {}).items():
                    tot[c] = tot.get(c, 0.0) + v
            summ = summarise(tot)
        else:
            c = dict(raw["counters"][key])
            if group:  # per step = `group` launches
                c = {k: (v * group if k.startswith(("SQ_", "FETCH", "WRITE", "ns")) else v) for k, v in c.items()}
            summ = summarise(c)
        doc = dict(workload=workload, kernel=what, kernel_source_sha256_16=bench.kernel_source_hash(which),
                   source=f"rocprofv3 --pmc passes (tools/gpu/pmc_passes.py: a few counters per pass, FETCH_SIZE and WRITE_SIZE in passes of their own) "
                          f"of `{raw['command']}`; FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md, WRITE_SIZE (KB) as is; raw: profiles/{tag}_{raw_name}",
                   **summ)
        if units:
            doc["units_in_profile"] = units
        json.dump(doc, open(os.path.join(ROOT, "profiles", out_name), "w"), indent=1)
        json.dump(raw, open(os.path.join(ROOT, "profiles", f"{tag}_{raw_name}"), "w"), indent=1)
        print(out_name, {k: v for k, v in doc.items() if k not in ("raw", "source")})


if __name__ == "__main__":
    main()
