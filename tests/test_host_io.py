"""Host-side file plumbing of the tile loop (round 6): the writer pool (utils/async_io.py), the float32 PLY fast path
(utils/ply.read_xyz32) and the text writers' formatting split over cores (f4l_write_rows_txt / f4l_write_partition_txt: the same
bytes whatever the number of workers).  No GPU."""
import ctypes as C
import os
import threading
import time

import numpy as np
import pytest

from fusion4landslide_amd.utils import async_io
from fusion4landslide_amd.utils.ply import read_ply, read_xyz32, write_ply


def test_writer_pool_runs_jobs_and_raises_their_errors(tmp_path, monkeypatch):
    monkeypatch.delenv("F4L_ASYNC_IO", raising=False)
    assert async_io.enabled()
    done, main = [], threading.get_ident()

    def job(i, fail=False):
        time.sleep(0.01)
        if fail:
            raise RuntimeError(f"job {i} failed")
        done.append((i, threading.get_ident()))
        return i
    futs = [async_io.submit(job, i) for i in range(8)]
    async_io.drain()
    assert sorted(i for i, _ in done) == list(range(8)) and all(t != main for _, t in done)   # on writer threads, all through
    assert [f.result() for f in futs] == list(range(8))
    async_io.drain()                                                                         # nothing pending: a no-op
    async_io.submit(job, 100)
    async_io.submit(job, 101, fail=True)
    async_io.submit(job, 102)
    with pytest.raises(RuntimeError, match="job 101"):
        async_io.drain()                                                                     # ... after every job was waited for
    assert {100, 102} <= {i for i, _ in done}
    async_io.drain()                                                                         # (the failure is reported once)
    # the serial order of work: every job runs where it is submitted
    monkeypatch.setenv("F4L_ASYNC_IO", "0")
    before = len(done)
    f = async_io.submit(job, 200)
    assert f.result() == 200 and done[-1] == (200, main) and len(done) == before + 1
    with pytest.raises(RuntimeError):
        async_io.submit(job, 201, fail=True)


def test_prefetch_reads_ahead_and_take_falls_back(tmp_path, monkeypatch):
    monkeypatch.delenv("F4L_ASYNC_IO", raising=False)
    rng = np.random.default_rng(0)
    a = rng.normal(size=(5000, 3)).astype(np.float32)
    p = str(tmp_path / "a.ply")
    write_ply(p, a)
    calls = []

    def reader(path):
        calls.append(path)
        return read_xyz32(path)
    async_io.prefetch(p, reader)
    async_io.prefetch(p, reader)                    # (a second request for the same file does not read it twice)
    assert np.array_equal(async_io.take(p, reader), a) and calls == [p]
    assert np.array_equal(async_io.take(p, reader), a) and calls == [p, p]   # nothing started: read now
    async_io.prefetch(str(tmp_path / "missing.ply"), reader)
    async_io.forget_prefetched()                    # a read that was started and never taken (and failed) is dropped quietly


def test_read_xyz32_is_read_ply_in_float32(tmp_path):
    rng = np.random.default_rng(1)
    a = (rng.normal(size=(3000, 3)) * [100, 100, 5] + [2600000, 1200000, 1500]).astype(np.float32)
    paths = {}
    paths["f32"] = str(tmp_path / "f32.ply"); write_ply(paths["f32"], a)
    paths["f64"] = str(tmp_path / "f64.ply"); write_ply(paths["f64"], a, dtype="float64")
    paths["rgb"] = str(tmp_path / "rgb.ply")
    rec = np.empty(len(a), dtype=[("red", "u1"), ("x", "<f4"), ("green", "u1"), ("y", "<f4"), ("z", "<f4"), ("blue", "u1")])  # any property order
    rec["x"], rec["y"], rec["z"] = a[:, 0], a[:, 1], a[:, 2]
    rec["red"] = rec["green"] = rec["blue"] = 7
    with open(paths["rgb"], "wb") as f:
        f.write((f"ply\nformat binary_little_endian 1.0\nelement vertex {len(a)}\nproperty uchar red\nproperty float x\nproperty uchar green\n"
                 "property float y\nproperty float z\nproperty uchar blue\nend_header\n").encode())
        f.write(rec.tobytes())
    paths["big"] = str(tmp_path / "big.ply")
    with open(paths["big"], "wb") as f:
        f.write(f"ply\nformat binary_big_endian 1.0\nelement vertex {len(a)}\nproperty float x\nproperty float y\nproperty float z\nend_header\n".encode())
        f.write(a.astype(">f4").tobytes())
    paths["ascii"] = str(tmp_path / "ascii.ply")
    with open(paths["ascii"], "w") as f:
        f.write(f"ply\nformat ascii 1.0\nelement vertex {len(a)}\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
        np.savetxt(f, a, fmt="%.9g")
    for kind, p in paths.items():
        got = read_xyz32(p)
        assert got.dtype == np.float32 and got.flags["C_CONTIGUOUS"], kind
        assert np.array_equal(got, np.ascontiguousarray(read_ply(p)[0], dtype=np.float32)), kind
        assert np.array_equal(got, a), kind
    with open(tmp_path / "not.ply", "w") as f:
        f.write("hello\n")
    with pytest.raises(ValueError):
        read_xyz32(str(tmp_path / "not.ply"))


def test_text_writers_write_the_same_bytes_on_any_number_of_cores(tmp_path, monkeypatch):
    from fusion4landslide_amd._lib import check, lib
    rng = np.random.default_rng(2)
    rows = (rng.normal(size=(70_001, 6)) * 10.0 ** rng.integers(-6, 5, (70_001, 1))).astype(np.float32)  # (more than four blocks of 16384 rows, a ragged last one)
    xyz = np.ascontiguousarray(rows[:, :3])
    lab = rng.integers(0, 300, len(rows)).astype(np.int32)
    out = {}
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("F4L_WRITER_THREADS", threads)
        pr, pp = tmp_path / f"rows_{threads}.txt", tmp_path / f"part_{threads}.txt"
        check(lib().f4l_write_rows_txt(str(pr).encode(), rows.ctypes.data_as(C.c_void_p), rows.shape[0], 6), "f4l_write_rows_txt")
        check(lib().f4l_write_partition_txt(str(pp).encode(), xyz.ctypes.data_as(C.c_void_p), lab.ctypes.data_as(C.c_void_p), len(lab), 300), "f4l_write_partition_txt")
        out[threads] = (open(pr, "rb").read(), open(pp, "rb").read())
    assert out["1"] == out["3"] == out["8"]
    ref = tmp_path / "ref.txt"
    np.savetxt(ref, rows, delimiter=" ", fmt="%.6f")
    assert out["8"][0] == open(ref, "rb").read()
    assert out["8"][1].count(b"\n") == len(rows) and out["8"][1].split(b"\n")[5].split()[6] == str(lab[5]).encode()
    # a label outside [0, n_supervoxels) is refused whichever block it sits in
    bad = lab.copy(); bad[60_000] = 300
    assert lib().f4l_write_partition_txt(str(tmp_path / "bad.txt").encode(), xyz.ctypes.data_as(C.c_void_p), bad.ctypes.data_as(C.c_void_p), len(bad), 300) != 0
