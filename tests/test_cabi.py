"""CPU suite, part 2: the C-ABI library loads and exports every symbol include/f4l.h declares (no compute)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def _declared():
    text = open(os.path.join(ROOT, "include", "f4l.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(f4l_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    names = _declared()
    for must in ["f4l_kabsch_batched", "f4l_piecewise_icp", "f4l_supervoxel", "f4l_knn", "f4l_normals",
                 "f4l_apply_transform", "f4l_nn_refine", "f4l_patch_normals"]:
        assert must in names


def test_library_exports_every_declared_symbol():
    from fusion4landslide_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail(f"{_lib.LIB_PATH} not built: run __graft_entry__.build()")
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(handle, name), f"{name} declared in include/f4l.h but not exported"
    assert set(_declared()) == set(_lib.SIGNATURES), "python binding table and header drifted apart"
    lib = _lib.lib()
    assert lib.f4l_version() >= 100
    assert lib.f4l_strerror(-1) == b"invalid argument"
    # pure-host entry points may be exercised without a GPU
    assert lib.f4l_knn_workspace_bytes(0, 30) == 0


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "fusion4landslide_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in src and "from oracle" not in src and "libf4l_oracle" not in src, f


def test_compute_entry_points_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from fusion4landslide_amd import engine
    from fusion4landslide_amd._lib import F4LError
    x = torch.zeros((4, 3))
    off = torch.tensor([0, 4])
    with pytest.raises(F4LError):
        engine.kabsch_batched(x, x, off)
    with pytest.raises(F4LError):
        engine.piecewise_icp(x, off, x, off)
    with pytest.raises(F4LError):
        engine.knn(x, 2)


def test_host_segmentation_matches_oracle_and_golden(golden_dir):
    """f4l_supervoxel_segment_host is host-only C++ inside the product library: check it on CPU against the
    golden labels (reference-generated) given the golden neighbours/normals."""
    import numpy as np
    from fusion4landslide_amd import _lib
    lib = _lib.lib()
    for name in ["surf_s0_n2000_k15", "georef_s3_n3000_k30", "surf_s4_n20000_k30"]:
        g = np.load(os.path.join(golden_dir, f"supervoxel_{name}.npz"))
        xyz = np.ascontiguousarray(g["xyz"], np.float32)
        nrm = np.ascontiguousarray(g["normals"], np.float64)
        knn = np.ascontiguousarray(g["knn_idx"], np.int32)
        labels = np.empty(len(xyz), np.int32)
        nsv = ctypes.c_int32(0)
        rc = lib.f4l_supervoxel_segment_host(xyz.ctypes.data, nrm.ctypes.data, knn.ctypes.data, len(xyz), int(g["k"]),
                                             float(g["resolution"]), labels.ctypes.data, ctypes.byref(nsv))
        assert rc == 0
        assert nsv.value == int(g["n_supervoxels"])
        assert np.array_equal(labels, g["labels"])


def test_reference_module_layout_is_importable():
    """The mirrors of the reference's modules import without a GPU (compute raises, importing does not)."""
    import importlib
    for mod in ["fusion4landslide_amd.cpp_core.supervoxel_segmentation.build.supervoxel",
                "fusion4landslide_amd.utils.o3d_tools", "fusion4landslide_amd.scripts.weighted_svd",
                "fusion4landslide_amd.src.piecewise_icp", "fusion4landslide_amd.main_piecewise_icp",
                "fusion4landslide_amd.src.functions", "fusion4landslide_amd.cpp_core.pcd_tiling.build.pcd_tiling"]:
        importlib.import_module(mod)
    pt = importlib.import_module("fusion4landslide_amd.cpp_core.pcd_tiling.build.pcd_tiling")
    assert callable(pt.tile_point_clouds) and callable(pt.resave_point_cloud)
    sv = importlib.import_module("fusion4landslide_amd.cpp_core.supervoxel_segmentation.build.supervoxel")
    assert callable(sv.computeSupervoxel) and callable(sv.WritePoints)
