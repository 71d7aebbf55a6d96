"""Measurement (library built with -DSVX_MEASURE_PREFIX): per pass of the exact segmentation, the lowest centre whose output changed.
Usage: F4L_LIB_PATH=.../lib_svx_prefix.so python tools/gpu/svx_prefix.py [n]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = synthetic.make_patches_device(n, int(round(45 * (n / 1e6) ** 0.5)), 1.386, torch.device("cuda"), seed=0)
lab, K = engine.supervoxel(d["src"], 30, 1.386 if n <= 1_000_000 else 0.52)
print("K", K)
