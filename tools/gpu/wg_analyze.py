import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import synthetic
wg = np.fromfile("gpurun_out/wg.bin", dtype=np.uint64).reshape(-1, 2).astype(np.int64)
d = synthetic.make_patches(1_000_000, 45, 1.386, seed=0)
ns, nt = np.diff(d["src_off"]), np.diff(d["tgt_off"])
start = (wg[:, 0] - wg[:, 0].min()) * 0.01
dur = (wg[:, 1] - wg[:, 0]) * 0.01
print("corr dur~ns", np.corrcoef(dur, ns)[0, 1], "dur~nt", np.corrcoef(dur, nt)[0, 1], "dur~start", np.corrcoef(dur, start)[0, 1])
first = start < 5
print("first round: n", first.sum(), "dur mean/min/max", dur[first].mean(), dur[first].min(), dur[first].max())
print("second round: n", (~first).sum(), "dur mean/min/max", dur[~first].mean(), dur[~first].min(), dur[~first].max())
print("percentiles dur first:", np.percentile(dur[first], [5, 25, 50, 75, 95]))
print("percentiles dur second:", np.percentile(dur[~first], [5, 25, 50, 75, 95]))
print("percentiles start second:", np.percentile(start[~first], [5, 25, 50, 75, 95]))
# block id order vs start
idx = np.arange(len(dur))
print("corr start~idx", np.corrcoef(start, idx)[0, 1])
for lo in range(0, 2025, 256):
    sel = slice(lo, lo + 256)
    print(lo, "start mean %.0f dur mean %.0f ns mean %.0f" % (start[sel].mean(), dur[sel].mean(), ns[sel].mean()))
