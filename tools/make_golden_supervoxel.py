#!/usr/bin/env python3
"""Generate tests/golden/supervoxel_*.npz by RUNNING the reference's own codelibrary templates.

Runs only in the build container: needs oracle/_ref/libf4l_ref.so, which oracle/Makefile compiles from
the headers where they lie under /root/reference/cpp_core/supervoxel_segmentation (see
oracle/ref_harness.cpp).  Output is data only: the float32 input cloud and what the reference computed
for it (kNN indices / squared distances, PCA normals, #grid cells, the fusion's starting lambda, #supervoxels, labels).
"""
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def surface_cloud(seed, n, extent=1.0, noise=0.002):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(0, extent, (n, 2))
    z = 0.1 * np.sin(6 * xy[:, 0] / extent) * np.cos(5 * xy[:, 1] / extent) + 0.03 * np.sin(23 * xy[:, 0] / extent)
    z = z + rng.normal(0, noise, n)
    return np.c_[xy, z].astype(np.float32)


def volume_cloud(seed, n):
    rng = np.random.default_rng(seed)
    return rng.uniform(-1, 1, (n, 3)).astype(np.float32)


def step_cloud(seed, n):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(0, 1, (n, 2))
    z = np.where(xy[:, 0] > 0.5, 0.08, 0.0) + rng.normal(0, 0.002, n)
    return np.c_[xy, z].astype(np.float32)


def slab_cloud(seed, n):
    rng = np.random.default_rng(seed)
    return (rng.uniform(0, 1, (n, 3)) * np.array([2.0, 2.0, 0.3])).astype(np.float32)


def lattice_cloud(m):
    g = np.arange(m, dtype=np.float32) * np.float32(0.125)  # exactly representable -> exact distance ties
    x, y = np.meshgrid(g, g, indexing="ij")
    return np.stack([x.ravel(), y.ravel(), np.zeros(m * m, np.float32)], axis=1)


def main():
    if not O.have_ref():
        raise SystemExit("oracle/_ref/libf4l_ref.so missing: run `make -C oracle` in the build container")
    out_dir = os.path.join(ROOT, "tests", "golden")
    cases = [
        ("surf_s0_n2000_k15", surface_cloud(0, 2000), 15, 0.15, True),
        ("surf_s1_n2000_k30", surface_cloud(1, 2000), 30, 0.2, True),
        ("vol_s2_n2000_k15", volume_cloud(2, 2000), 15, 0.5, True),
        ("georef_s3_n3000_k30", surface_cloud(3, 3000, extent=20.0, noise=0.01) + np.array([2647.0, 1177.0, 1500.0], np.float32), 30, 2.5, True),
        ("lattice_m24_k9", lattice_cloud(24), 9, 0.5, True),
        ("surf_s4_n20000_k30", surface_cloud(4, 20000), 30, 0.1, False),
        # round 3: a step between two levels (the normal term of the metric decides along the edge) and a thin slab of volume
        ("step_s5_n4000_k12", step_cloud(5, 4000), 12, 0.12, False),
        ("slab_s6_n3000_k20", slab_cloud(6, 3000), 20, 0.35, False),
    ]
    for name, xyz, k, res, with_d2 in cases:
        r = O.ref_supervoxel(xyz, k, res)
        arrays = dict(xyz=xyz, k=np.int32(k), resolution=np.float64(res), knn_idx=r["knn_idx"],
                      normals=r["normals"], labels=r["labels"], n_supervoxels=np.int32(r["n_supervoxels"]),
                      n_grid_cells=np.int32(r["n_grid_cells"]),
                      lambda0=np.float64(O.ref_lambda0(xyz, r["normals"], r["knn_idx"], res)))
        if with_d2:
            arrays["knn_d2"] = r["knn_d2"]
        path = os.path.join(out_dir, f"supervoxel_{name}.npz")
        np.savez_compressed(path, **arrays)
        print(f"{name}: n={xyz.shape[0]} k={k} res={res} K={r['n_supervoxels']} cells={r['n_grid_cells']} lambda0={float(arrays['lambda0'])!r} "
              f"-> {os.path.getsize(path)} bytes")


def large_fixture():
    """tests/golden/sv_large_ref.npz: what the reference's own templates compute on a 300 k-point cloud -- 15 x the largest
    of the small fixtures -- WITHOUT the cloud and its neighbour lists (regenerated from the seed at test time,
    tests/_util.py::large_surface_cloud): labels, K, the grid-cell count, lambda0, and checksums of the cloud's bits, of the
    neighbour indices and of the squared distances' bits."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _util import LARGE_CASE, bits_checksum, large_surface_cloud
    c = LARGE_CASE
    xyz = large_surface_cloud(c["seed"], c["n"], c["extent"])
    r = O.ref_supervoxel(xyz, c["k"], c["resolution"])
    lam = O.ref_lambda0(xyz, r["normals"], r["knn_idx"], c["resolution"])
    out = os.path.join(ROOT, "tests", "golden", "sv_large_ref.npz")
    np.savez_compressed(out, labels=r["labels"].astype(np.int32), n_supervoxels=np.int32(r["n_supervoxels"]),
                        n_grid_cells=np.int32(r["n_grid_cells"]), lambda0=np.float64(lam),
                        xyz_checksum=np.uint64(bits_checksum(xyz)), knn_idx_checksum=np.uint64(bits_checksum(r["knn_idx"])),
                        knn_d2_checksum=np.uint64(bits_checksum(r["knn_d2"])),
                        normals_abs_sum=np.abs(r["normals"]).sum(axis=0))
    print(f"large: n={c['n']} k={c['k']} res={c['resolution']} K={r['n_supervoxels']} cells={r['n_grid_cells']} lambda0={lam!r} "
          f"-> {os.path.getsize(out)} bytes")


def partition_text_fixture():
    """tests/golden/partition_txt_ref.npz: a small labelled cloud and the bytes the reference's own writer
    (codelibrary/geometry/io/xyz_io.h:192-221, driven like supervoxel.cpp:45-64) puts into the partition text file for it:
    coordinates of every magnitude the 12-significant-digit formatting treats differently (georeferenced, sub-millimetre,
    negative, exact integers, values that print in scientific notation)."""
    import tempfile
    rng = np.random.default_rng(7)
    xyz = np.concatenate([
        rng.uniform(-5, 5, (60, 3)), rng.uniform(0, 1, (40, 3)) + [2647123.0, 1177456.0, 1500.0], rng.uniform(-1e-4, 1e-4, (30, 3)),
        np.array([[0.0, 1.0, -2.0], [1e-7, -3e-9, 5e10], [123456.789, 0.5, 0.25], [1e15, 3.0, -0.0]]), rng.normal(0, 1e3, (40, 3))]).astype(np.float32)
    K = 23
    labels = rng.integers(0, K, len(xyz)).astype(np.int32)
    labels[:K] = np.arange(K)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "ref.txt")
        O.ref_write_points(path, K, xyz, labels)
        text = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
    out = os.path.join(ROOT, "tests", "golden", "partition_txt_ref.npz")
    np.savez_compressed(out, xyz=xyz, labels=labels, n_supervoxels=np.int32(K), text=text)
    print(f"partition_txt_ref: {len(xyz)} points, {len(text)} bytes of text -> {os.path.getsize(out)} bytes")


if __name__ == "__main__":
    if "--large" in sys.argv:
        large_fixture()
    else:
        if "--text-only" not in sys.argv:
            main()
        partition_text_fixture()
