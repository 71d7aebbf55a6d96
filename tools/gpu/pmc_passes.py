"""Counter passes of one command under rocprofv3, summarised per kernel (run on the GPU box, from the repo root):

    python3 tools/gpu/pmc_passes.py <out.json> <kernel substring>[,<substring>...] -- python3 bench.py --config ... --steps 3

Every pass is its own `rocprofv3 --pmc <a few counters> --kernel-trace` run of the command (MI355X_MICROARCH.md: FETCH_SIZE and
WRITE_SIZE do not fit one pass; a pass that asks for more counters of one block than the hardware has aborts inside rocprofv3
and never returns -- hence few counters per pass and a time limit on each).  This driver never touches the GPU itself; the
program rocprofv3 starts is the one after `--`.  Output: {kernel substring: {counter: mean per launch, "launches": n,
"ns": mean duration, "per_call": {...}}}, counters in their own units (FETCH_SIZE / WRITE_SIZE: KB).

`--counters "A B;C D"`: passes of one's own instead of the built-in list (an unknown counter name fails that pass only).
`--group N`: the command issues the kernel N times per logical call (e.g. two icp_kernel launches per f4l_patch_loop step, or the
~270 launches of one segmentation); "per_call" sums N consecutive launches.  `--sum-all`: per_call = the sum over ALL kernels
whose name contains the substring, divided by --calls (for a family like `svg::`)."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

PASSES = [
    "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY",
    "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64",
    "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32",
    "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_LDS",
    "FETCH_SIZE",
    "WRITE_SIZE",
]


def main():
    argv = sys.argv[1:]
    calls, sum_all, passes = 1, False, PASSES
    while argv and argv[0].startswith("--") and argv[0] != "--":
        if argv[0] == "--calls":
            calls = int(argv[1]); argv = argv[2:]
        elif argv[0] == "--sum-all":
            sum_all = True; argv = argv[1:]
        elif argv[0] == "--passes":
            passes = [PASSES[int(i)] for i in argv[1].split(",")]; argv = argv[2:]
        elif argv[0] == "--counters":  # passes of one's own: "A B;C D" = two passes
            passes = [c.strip() for c in argv[1].split(";") if c.strip()]; argv = argv[2:]
        else:
            raise SystemExit("unknown option " + argv[0])
    out_path, names = argv[0], argv[1].split(",")
    cmd = argv[argv.index("--") + 1:]
    root = os.getcwd()
    tmp = os.path.join(root, "gpurun_out", "pmc_tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    res = {k: collections.defaultdict(float) for k in names}
    launches = {k: 0 for k in names}
    by_kernel = collections.defaultdict(lambda: collections.defaultdict(float))  # short kernel name -> counter -> sum over the run
    for i, counters in enumerate(passes):
        shutil.rmtree(tmp, ignore_errors=True)
        full = ["rocprofv3", "--pmc"] + counters.split() + ["--kernel-trace", "--output-format", "csv", "-d", tmp, "--"] + cmd
        try:
            rc = subprocess.run(full, env=env, cwd=root, timeout=240, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode
        except subprocess.TimeoutExpired:
            print(f"pass {i} ({counters}): TIMEOUT", flush=True)
            continue
        fs = glob.glob(os.path.join(tmp, "*", "*_counter_collection.csv"))
        print(f"pass {i} ({counters}): rc={rc} files={len(fs)}", flush=True)
        if not fs:
            continue
        per = {k: collections.defaultdict(lambda: collections.defaultdict(float)) for k in names}  # name -> dispatch -> counter -> value
        dur = {k: {} for k in names}
        for row in csv.DictReader(open(fs[0])):
            for k in names:
                if k in row["Kernel_Name"]:
                    per[k][row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"])
                    dur[k][row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                    short = row["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
                    by_kernel[short][row["Counter_Name"]] += float(row["Counter_Value"])
        for k in names:
            n = len(per[k])
            if not n:
                continue
            launches[k] = n
            tot = collections.defaultdict(float)
            for d in per[k].values():
                for c, v in d.items():
                    tot[c] += v
            for c, v in tot.items():
                res[k][c] = v / n
            res[k]["ns"] = sum(dur[k].values()) / n
            res[k]["ns_total"] = float(sum(dur[k].values()))
    shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for k in names:
        d = dict(res[k])
        d["launches"] = launches[k]
        if sum_all and launches[k]:
            d["per_call"] = {c: v * launches[k] / calls for c, v in res[k].items() if c not in ("ns_total",)}
            d["calls"] = calls
        out[k] = d
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    top = {k: {c: v / calls for c, v in d.items()} for k, d in by_kernel.items()} if sum_all else {}
    json.dump(dict(command=" ".join(cmd), counters=out, per_call_by_kernel=top), open(out_path, "w"), indent=1)
    print(json.dumps(out)[:3000])


if __name__ == "__main__":
    main()
