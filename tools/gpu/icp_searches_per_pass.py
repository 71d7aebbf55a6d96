"""How many source points are SEARCHED in each pass of icp_kernel (profiling build with the search counters: F4L_LIB_PATH at a
`make PROF=1` library): the launch is repeated with 0, 1, 2, ... iterations and the totals are differenced.
    F4L_LIB_PATH=$PWD/fusion4landslide_amd/lib/variants/lib_icp_prof.so python3 tools/gpu/icp_searches_per_pass.py [config]"""
import os, re, subprocess, sys
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4_50M_100k"
if len(sys.argv) > 2 and sys.argv[2] == "child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench
    from fusion4landslide_amd import engine, synthetic
    mi = int(sys.argv[3])
    os.environ["F4L_ICP_PROF"] = "1"; os.environ["F4L_ICP_DEBUG"] = "64"; os.environ["F4L_ICP_THROUGHPUT"] = "1"
    c = synthetic.CONFIGS[cfg]  # (the child used to hard-code C2_1M_2k whatever was asked for)
    d = synthetic.make_patches_device(c["n"], c["cells"], c["resolution"], torch.device("cuda"))
    prob = bench.Problem(torch, engine, synthetic, d, torch.device("cuda"))
    engine.patch_loop(d["src"], d["src_off"], d["tgt"], d["tgt_off"], prob.cs, prob.ct, prob.coff, None, 0.0, 1e-6, max_corr_dist=0.1,
                      max_iter=mi, fixed_iters=True, max_src_patch=d["max_src"], max_tgt_patch=d["max_tgt"], search="f64")
    torch.cuda.synchronize()
    sys.exit(0)
prev = 0.0
for mi in (0, 1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20):
    r = subprocess.run([sys.executable, __file__, cfg, "child", str(mi)], capture_output=True, text=True)
    m = re.search(r"searched ([0-9.]+) of ([0-9.]+), wave-batches ([0-9.]+)", r.stderr)
    q = re.search(r"per query: steps ([0-9.]+)", r.stderr)
    if not m:
        print(mi, "no counters", r.stderr[-300:]); continue
    tot = float(m.group(1)) * (mi + 1)
    print(f"{cfg} passes 0..{mi}: searched per patch in all {tot:8.1f} (largest patch {m.group(2)}), since the previous line {tot - prev:7.1f}; "
          f"steps per query so far {q.group(1) if q else '?'}")
    prev = tot
