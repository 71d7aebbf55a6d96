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


def test_partition_text_file_is_byte_identical_to_the_reference_writer(golden_dir, tmp_path):
    """f4l_write_partition_txt (host only, no GPU needed) against the bytes the reference's own header-only writer produced
    (xyz_io.h:192-221 driven like supervoxel.cpp:45-64; tests/golden/partition_txt_ref.npz from
    tools/make_golden_supervoxel.py) -- and, where the reference-backed harness is present (the build container), against
    that writer run now on a golden cloud."""
    import ctypes as C

    import numpy as np

    from fusion4landslide_amd._lib import lib

    def ours(path, xyz, labels, K):
        xyz, labels = np.ascontiguousarray(xyz, np.float32), np.ascontiguousarray(labels, np.int32)
        rc = lib().f4l_write_partition_txt(str(path).encode(), xyz.ctypes.data_as(C.c_void_p), labels.ctypes.data_as(C.c_void_p),
                                           len(xyz), int(K))
        assert rc == 0
        return open(path, "rb").read()

    g = np.load(os.path.join(golden_dir, "partition_txt_ref.npz"))
    assert ours(tmp_path / "a.txt", g["xyz"], g["labels"], g["n_supervoxels"]) == g["text"].tobytes()
    from oracle import oracle as O
    if O.have_ref() and hasattr(O.ref(), "f4l_ref_write_points"):
        c = np.load(os.path.join(golden_dir, "supervoxel_georef_s3_n3000_k30.npz"))
        O.ref_write_points(tmp_path / "ref.txt", int(c["n_supervoxels"]), c["xyz"], c["labels"])
        assert ours(tmp_path / "b.txt", c["xyz"], c["labels"], c["n_supervoxels"]) == open(tmp_path / "ref.txt", "rb").read()


def test_partition_text_coordinates_are_printf_12g(tmp_path):
    """The partition file's coordinates are `out << std::setprecision(12) << double(x)` in the reference (xyz_io.h:192-221) =
    printf("%.12g").  The writer forms the twelve digits itself for 1 <= |x| < 10^12 (round 5: 1.84 -> 0.23 s per million points):
    every coordinate must be the string Python's '%.12g' gives for the same double -- on magnitudes from 10^-6 to 10^13, values that
    round up into a new digit, trailing zeros, signed zeros, the float32 extremes."""
    import ctypes as C
    import numpy as np
    from fusion4landslide_amd._lib import check, lib
    rng = np.random.default_rng(11)
    n = 60_000
    pts = (rng.normal(size=(n, 3)) * 10.0 ** rng.integers(-6, 14, (n, 1))).astype(np.float32)
    edge = np.array([0.0, -0.0, 1.0, -1.0, 9.99999999999, 999999999999.6, 1e12, 0.5, 123456.789, 99999.9999999, 1e11, 2.5, 1 / 128,
                     16777216.0, 3.4e38, 1e-45, 9.9999998e11, 999999.94, 1000000.06, 7.0000005, 10.0, 100.0, 1e5, -1e5], np.float32)
    pts.reshape(-1)[:len(edge)] = edge
    lab = rng.integers(0, 700, n).astype(np.int32)
    path = tmp_path / "p.txt"
    check(lib().f4l_write_partition_txt(str(path).encode(), pts.ctypes.data_as(C.c_void_p), lab.ctypes.data_as(C.c_void_p), n, 700),
          "f4l_write_partition_txt")
    lines = open(path).read().split("\n")
    assert len(lines) == n + 1 and lines[-1] == ""
    for i in range(n):
        f = lines[i].split(" ")
        assert len(f) == 7 and f[:3] == ["%.12g" % float(v) for v in pts[i]] and int(f[6]) == lab[i], (i, lines[i])
    assert lib().f4l_write_partition_txt(str(path).encode(), pts.ctypes.data_as(C.c_void_p), lab.ctypes.data_as(C.c_void_p), n, 5) != 0  # label beyond K


def test_sanitizer_run_of_the_oracle_and_the_host_code():
    """`make -C oracle asan`: the C oracle and the product library's host-only code (the sequential segmentation and the
    text writer, csrc/supervoxel_host.cpp) under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU; both
    segmentations must agree and no report may fire."""
    import shutil
    import subprocess

    import pytest
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++") or shutil.which("make") is None:
        pytest.skip("no clang with sanitizer runtimes here")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "sanitize_check ok" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
