"""fusion4landslide_amd -- MI355X-native piecewise-ICP displacement-field engine.

A drop-in for ONE hot path of gseg-ethz/fusion4landslide (supervoxel partition -> per-patch weighted
Kabsch -> per-patch ICP -> displacement rows), written as hand-written HIP kernels for gfx950 behind a
C ABI (include/f4l.h, fusion4landslide_amd/lib/libf4l_hip.so).  The sub-packages mirror the reference's
own module layout so that the reference's call sites read the same:

    fusion4landslide_amd.cpp_core.supervoxel_segmentation.build.supervoxel   computeSupervoxel(...)
    fusion4landslide_amd.utils.o3d_tools                                     icp_registration(...)
    fusion4landslide_amd.scripts.weighted_svd                                weighted_procrustes(...)
    fusion4landslide_amd.src.piecewise_icp                                   Piecewise_ICP(cfg)

There is NO CPU fallback: importing is cheap, but every compute entry point raises if the HIP library or
a GPU is missing.
"""
from . import _lib  # noqa: F401
from .engine import (apply_transform, kabsch_batched, kabsch_residuals, patch_normals,  # noqa: F401
                     piecewise_icp)

__version__ = "0.1.0"
