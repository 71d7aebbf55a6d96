"""Tensor-level host API over the C ABI (include/f4l.h): torch tensors in, torch tensors out.

PyTorch is plumbing here (device memory, streams); every computation below is one call into
libf4l_hip.so on the current torch stream.  Inputs must live on the GPU; nothing falls back to the CPU.

Ragged patches are CSR: ``pts`` is (n, 3) float32 with the points of patch p in rows
``off[p]:off[p+1]`` (``off`` int64, length P+1).  This is the layout that replaces the reference's Python
lists of index tensors (src/coarse_to_fine_matching_base.py:3156-3157, 3254).
"""
import ctypes as C
import os
import threading

from . import _lib
from ._lib import check, lib, ptr, require_gpu, stream_ptr

_ICP_MODES = {"point2point": _lib.ICP_POINT2POINT, "point2plane": _lib.ICP_POINT2PLANE}


def _p2plane_bit(p2plane):
    if p2plane not in ("robust", "open3d"):
        raise ValueError("p2plane must be 'robust' or 'open3d'")
    return _lib.ICP_P2PL_OPEN3D if p2plane == "open3d" else 0


def _dev(t, dtype, name, shape_tail=None):
    torch = require_gpu()
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor on the GPU")
    if not t.is_cuda:
        raise _lib.F4LError(f"{name} must be a CUDA (ROCm) tensor; there is no CPU path")
    if t.dtype != dtype:
        t = t.to(dtype)
    if not t.is_contiguous():
        t = t.contiguous()
    if shape_tail is not None and tuple(t.shape[1:]) != tuple(shape_tail):
        raise ValueError(f"{name} must have shape (n, {', '.join(map(str, shape_tail))}), got {tuple(t.shape)}")
    return t


_SCRATCH = {}
_SCRATCH_LOCK = threading.Lock()


def _scratch(nbytes, device):
    """Workspace of a stateless call: one buffer per (device, stream), grown when a call needs more and otherwise reused -- the calls
    of a stream run in order, so the next one may overwrite what the last one left.  (Handing every workspace back to the caching
    allocator made a 100 M-point path stall for 0.2-0.8 s now and then: the allocator splits a freed 14 GB block for the small
    tensors that follow and has to get a new one from the driver for the next large request.)  F4L_NO_SCRATCH_CACHE=1: a fresh
    tensor per call; :func:`release_scratch` drops the buffers."""
    torch = require_gpu()
    nbytes = max(int(nbytes), 1)
    if os.environ.get("F4L_NO_SCRATCH_CACHE") or torch.cuda.is_current_stream_capturing():
        return torch.empty((nbytes,), dtype=torch.uint8, device=device)
    # (the host thread is part of the key: f4l_knn and f4l_epoch_join synchronise in mid-call with the GIL released, so two threads
    #  on one stream must not share a buffer -- ADVICE r5)
    key = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream, threading.get_ident())
    with _SCRATCH_LOCK:
        buf = _SCRATCH.get(key)
        if buf is None or buf.numel() < nbytes:
            _SCRATCH.pop(key, None)
            buf = None  # (free the old one first: the two need not coexist)
            buf = torch.empty((nbytes,), dtype=torch.uint8, device=device)
            _SCRATCH[key] = buf
    return buf


def release_scratch():
    """Drops the cached workspaces (see :func:`_scratch`): the entry points call it when a run is over, so that the peak workspace
    (about 14 GB at 100 M points) does not stay allocated for the life of the process."""
    with _SCRATCH_LOCK:
        _SCRATCH.clear()


def _max_patch(off):
    if off.numel() < 2:
        return 0
    return int((off[1:] - off[:-1]).max().item())


def kabsch_batched(src, ref, off, weights=None, weight_thresh=0.0, eps=1e-7):
    """Batched ragged weighted Kabsch (scripts/weighted_svd.py:58-129 per patch).

    src, ref: (n, 3) float32 or float64; off: (P+1,) int64; weights: (n,) or None.
    Returns R (P, 3, 3) float64, t (P, 3) float64.
    """
    torch = require_gpu()
    f64 = src.dtype == torch.float64
    dt = torch.float64 if f64 else torch.float32
    src = _dev(src, dt, "src", (3,))
    ref = _dev(ref, dt, "ref", (3,))
    off = _dev(off, torch.int64, "off")
    w = None if weights is None else _dev(weights, dt, "weights")
    n, P = src.shape[0], off.shape[0] - 1
    if ref.shape[0] != n or (w is not None and w.shape[0] != n):
        raise ValueError("src, ref and weights must have the same number of rows")
    R = torch.empty((P, 3, 3), dtype=torch.float64, device=src.device)
    t = torch.empty((P, 3), dtype=torch.float64, device=src.device)
    fn = lib().f4l_kabsch_batched_f64 if f64 else lib().f4l_kabsch_batched
    check(fn(ptr(src), ptr(ref), ptr(w), ptr(off), P, n, float(weight_thresh), float(eps), ptr(R), ptr(t),
             stream_ptr()), "f4l_kabsch_batched")
    return R, t


def kabsch_transforms(src, ref, off, weights=None, weight_thresh=0.0, eps=1e-7):
    """`weighted_procrustes(..., return_transform=True)` per patch (scripts/weighted_svd.py:115-120): (P, 4, 4) float64,
    ready to be the `init_T` of :func:`piecewise_icp` (src/coarse_to_fine_matching_base.py:3341-3360)."""
    torch = require_gpu()
    src = _dev(src, torch.float32, "src", (3,))
    ref = _dev(ref, torch.float32, "ref", (3,))
    off = _dev(off, torch.int64, "off")
    w = None if weights is None else _dev(weights, torch.float32, "weights")
    n, P = src.shape[0], off.shape[0] - 1
    if ref.shape[0] != n or (w is not None and w.shape[0] != n):
        raise ValueError("src, ref and weights must have the same number of rows")
    T = torch.empty((P, 4, 4), dtype=torch.float64, device=src.device)
    check(lib().f4l_kabsch_transforms(ptr(src), ptr(ref), ptr(w), ptr(off), P, n, float(weight_thresh), float(eps), ptr(T),
                                      stream_ptr()), "f4l_kabsch_transforms")
    return T


def kabsch2_batched(src, ref, off, weights=None, normalize_w=True, w_threshold=0.0, eps=1e-7):
    """Kabsch #2 (src/functions.py:12-85 `kabsch_transformation_estimation`) per ragged batch element.

    src, ref: (n, 3) float32 or float64; off: (P+1,) int64; weights: (n,) or None.
    Returns R (P, 3, 3) float64 (third column scaled by det(V U^T) like the reference), t (P, 3) float64."""
    torch = require_gpu()
    f64 = src.dtype == torch.float64
    dt = torch.float64 if f64 else torch.float32
    src = _dev(src, dt, "src", (3,))
    ref = _dev(ref, dt, "ref", (3,))
    off = _dev(off, torch.int64, "off")
    w = None if weights is None else _dev(weights, dt, "weights")
    n, P = src.shape[0], off.shape[0] - 1
    if ref.shape[0] != n or (w is not None and w.shape[0] != n):
        raise ValueError("src, ref and weights must have the same number of rows")
    R = torch.empty((P, 3, 3), dtype=torch.float64, device=src.device)
    t = torch.empty((P, 3), dtype=torch.float64, device=src.device)
    fn = lib().f4l_kabsch2_batched_f64 if f64 else lib().f4l_kabsch2_batched
    check(fn(ptr(src), ptr(ref), ptr(w), ptr(off), P, n, int(bool(normalize_w)), float(w_threshold), float(eps), ptr(R),
             ptr(t), stream_ptr()), "f4l_kabsch2_batched")
    return R, t


def _median_of_sqrt(d2):
    """Device scalar: numpy's median (mean of the middle pair) of sqrt(d2) -- the square root first: the mean of the middle pair
    is taken over distances."""
    torch = require_gpu()
    if d2.dim() != 1:
        raise ValueError("d2 must be a vector (a column of a distance table: pass its view, the stride is taken from it)")
    stride = int(d2.stride(0)) if d2.shape[0] > 1 else 1
    out = torch.empty((1,), dtype=torch.float64, device=d2.device)
    nbytes = lib().f4l_median_f64_workspace_bytes(d2.shape[0])
    ws = _scratch(nbytes, d2.device)
    check(lib().f4l_median_sqrt_f64(ptr(d2), d2.shape[0], stride, ptr(out), ptr(ws), C.c_size_t(nbytes), stream_ptr()), "f4l_median_sqrt_f64")
    return out


def median_resolution(src, tgt=None, src_nn1_d2=None, tgt_nn1_d2=None):
    """`_compute_median_resolution` (src/coarse_to_fine_matching_base.py:2716-2754): median distance of every point to
    its nearest other point (exact 2-NN on the GPU, f4l_knn), the larger of the two clouds' medians when `tgt` is given.
    `src_nn1_d2`: the squared nearest-neighbour distances of `src` when a neighbour search of it has run already
    (`knn_normals(..., return_nn1=True)`): its 2-NN pass is skipped; `tgt_nn1_d2` likewise (`epoch_join`).  Returns a Python float."""
    torch = require_gpu()

    def one(xyz):
        _, d2 = knn(xyz, 2, return_d2=True)
        return _median_of_sqrt(d2[:, 1])

    r = one(src) if src_nn1_d2 is None else _median_of_sqrt(_dev(src_nn1_d2, torch.float64, "src_nn1_d2"))
    if tgt is None:
        return float(r.item())
    rt = one(tgt) if tgt_nn1_d2 is None else _median_of_sqrt(_dev(tgt_nn1_d2, torch.float64, "tgt_nn1_d2"))  # (epoch_join's)
    return float(torch.maximum(r, rt).item())  # one read-back for both clouds


def kabsch_residuals(src, ref, off, R, t):
    """|| R_p s_i + t_p - r_i || per row (scripts/weighted_svd.py:143-146) -> (n,) float64."""
    torch = require_gpu()
    src = _dev(src, torch.float32, "src", (3,))
    ref = _dev(ref, torch.float32, "ref", (3,))
    off = _dev(off, torch.int64, "off")
    R = _dev(R, torch.float64, "R")
    t = _dev(t, torch.float64, "t")
    n, P = src.shape[0], off.shape[0] - 1
    res = torch.empty((n,), dtype=torch.float64, device=src.device)
    check(lib().f4l_kabsch_residuals(ptr(src), ptr(ref), ptr(off), P, n, ptr(R), ptr(t), ptr(res), stream_ptr()),
          "f4l_kabsch_residuals")
    return res


def patch_normals(pts, off, knn=30, max_patch=None, f64=False):
    """Per-patch `estimate_normals()` (utils/o3d_tools.py:29-30) -> (n, 3) float32, or float64 with f64=True (what Open3D keeps
    and `registration_icp` reads: the point-to-plane launches take either)."""
    torch = require_gpu()
    pts = _dev(pts, torch.float32, "pts", (3,))
    off = _dev(off, torch.int64, "off")
    P = off.shape[0] - 1
    if max_patch is None:
        max_patch = _max_patch(off)
    out = torch.empty(pts.shape, dtype=torch.float64 if f64 else torch.float32, device=pts.device)
    fn = lib().f4l_patch_normals_f64 if f64 else lib().f4l_patch_normals
    check(fn(ptr(pts), ptr(off), P, int(knn), int(max_patch), ptr(out), stream_ptr()), "f4l_patch_normals")
    return out


def _target_normals(torch, tgt, tgt_off, max_tgt_patch, tgt_normals):
    """The target normals of a point-to-plane launch and the mode bit that says what they are: the caller's float32 or float64
    tensor, or -- none given -- float64 normals made here, like the doubles Open3D's `estimate_normals()` leaves in the cloud."""
    if tgt_normals is None:
        return patch_normals(tgt, tgt_off, 30, max_tgt_patch, f64=True), 0x200
    if tgt_normals.dtype == torch.float64:
        return _dev(tgt_normals, torch.float64, "tgt_normals", (3,)), 0x200
    return _dev(tgt_normals, torch.float32, "tgt_normals", (3,)), 0


def piecewise_icp(src, src_off, tgt, tgt_off, init_T=None, max_corr_dist=0.1, max_iter=30, rel_fitness=1e-6,
                  rel_rmse=1e-6, icp_type="point2point", fixed_iters=False, tgt_normals=None, return_corr=False,
                  max_src_patch=None, max_tgt_patch=None, search="f64", p2plane="robust", src_normals=None, gicp_epsilon=0.0):
    """Batched per-patch ICP (utils/o3d_tools.py:12-71 for P patch pairs in one launch).

    Returns dict(T (P,4,4) f64, fitness (P,) f64, rmse (P,) f64, iters (P,) i32[, corr (n_src,) i32]).
    ``fixed_iters=True`` is the benchmark mode (exactly ``max_iter`` updates, no early exit).
    ``search="f64"`` (default) evaluates positions, distances and sums in double like the reference's Open3D path
    and reproduces the CPU oracle to 1e-9 m; ``"f32"`` is the fast mode (float32 search on patch-relative coordinates,
    ~1.5x faster): same answer to ~1e-8 m on well-posed patches, but an ill-posed patch (displaced beyond the radius,
    low fitness) can end in a different local solution.
    ``p2plane`` (point-to-plane only): ``"robust"`` (default of the batched calls) leaves out a step whose 6 x 6 system does not
    pin its six unknowns (fewer than six pairs, singular to 1e-13); ``"open3d"`` is Open3D's own semantics -- Eigen's pivoted
    L D L^T in the caller's frame, applied whenever there is one correspondence (F4L_ICP_P2PL_OPEN3D, include/f4l.h) -- what
    the mirror of ``icp_registration`` asks for.  The same minimiser wherever the system is well posed.
    ``icp_type="generalized_icp"`` (utils/o3d_tools.py:40-41,51-56; f4l_piecewise_gicp): both patches' normals (float64, made here
    like `estimate_normals()` unless given) become Open3D's covariances; ``gicp_epsilon`` is the estimator's first parameter --
    the reference passes ``False`` = 0.0, Open3D's default is 1e-3.  Always the float64 search and Open3D's step semantics.
    """
    torch = require_gpu()
    generalized = icp_type == "generalized_icp"
    if icp_type not in _ICP_MODES and not generalized:
        raise ValueError("ICP type not supported")  # utils/o3d_tools.py:43
    mode = 0 if generalized else _ICP_MODES[icp_type] | _p2plane_bit(p2plane)
    src = _dev(src, torch.float32, "src", (3,))
    tgt = _dev(tgt, torch.float32, "tgt", (3,))
    src_off = _dev(src_off, torch.int64, "src_off")
    tgt_off = _dev(tgt_off, torch.int64, "tgt_off")
    P = src_off.shape[0] - 1
    if tgt_off.shape[0] - 1 != P:
        raise ValueError("src_off and tgt_off must describe the same number of patches")
    if max_src_patch is None:
        max_src_patch = _max_patch(src_off)
    if max_tgt_patch is None:
        max_tgt_patch = _max_patch(tgt_off)
    T0 = None
    if init_T is not None:
        T0 = _dev(init_T, torch.float64, "init_T")
        if T0.numel() != P * 16:
            raise ValueError("init_T must be (P, 4, 4)")
    tn, nbit = None, 0
    if icp_type == "point2plane":
        tn, nbit = _target_normals(torch, tgt, tgt_off, max_tgt_patch, tgt_normals)
    dev = src.device
    T = torch.empty((P, 4, 4), dtype=torch.float64, device=dev)
    fit = torch.empty((P,), dtype=torch.float64, device=dev)
    rmse = torch.empty((P,), dtype=torch.float64, device=dev)
    iters = torch.empty((P,), dtype=torch.int32, device=dev)
    corr = torch.empty((src.shape[0],), dtype=torch.int32, device=dev) if return_corr else None
    if generalized:
        sn = (patch_normals(src, src_off, 30, max_src_patch, f64=True) if src_normals is None
              else _dev(src_normals, torch.float64, "src_normals", (3,)))
        tn = (patch_normals(tgt, tgt_off, 30, max_tgt_patch, f64=True) if tgt_normals is None
              else _dev(tgt_normals, torch.float64, "tgt_normals", (3,)))
        if sn.shape[0] != src.shape[0] or tn.shape[0] != tgt.shape[0]:
            raise ValueError("one normal per point")
        check(lib().f4l_piecewise_gicp(ptr(src), ptr(src_off), ptr(tgt), ptr(tgt_off), P, ptr(T0), ptr(sn), ptr(tn),
                                       float(gicp_epsilon), float(max_corr_dist), int(max_iter), float(rel_fitness),
                                       float(rel_rmse), int(bool(fixed_iters)), int(max_src_patch), int(max_tgt_patch),
                                       int(src.shape[0]), ptr(T), ptr(fit), ptr(rmse), ptr(iters), ptr(corr), stream_ptr()),
              "f4l_piecewise_gicp")
        out = dict(T=T, fitness=fit, rmse=rmse, iters=iters)
        if return_corr:
            out["corr"] = corr
        return out
    check(lib().f4l_piecewise_icp(ptr(src), ptr(src_off), ptr(tgt), ptr(tgt_off), P, ptr(T0), ptr(tn),
                                  float(max_corr_dist), int(max_iter), float(rel_fitness), float(rel_rmse), mode | nbit,
                                  int(bool(fixed_iters)), {"f32": _lib.SEARCH_F32, "f64": _lib.SEARCH_F64}[search],
                                  int(max_src_patch), int(max_tgt_patch), int(src.shape[0]), ptr(T), ptr(fit),
                                  ptr(rmse), ptr(iters), ptr(corr), stream_ptr()), "f4l_piecewise_icp")
    out = dict(T=T, fitness=fit, rmse=rmse, iters=iters)
    if return_corr:
        out["corr"] = corr
    return out


def patch_loop(src, src_off, tgt, tgt_off, corr_src, corr_ref, corr_off, corr_weights=None, weight_thresh=0.0, eps=1e-6,
               max_corr_dist=0.1, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6, icp_type="point2point", fixed_iters=False,
               tgt_normals=None, return_corr=False, return_rows=True, max_src_patch=None, max_tgt_patch=None, search="f64",
               rows_src=None, rows_off=None, min_corr=0, init_round_f32=False, p2plane="robust"):
    """The per-patch loop body of src/coarse_to_fine_matching_base.py:3338-3408 in one launch (f4l_patch_loop):
    weighted Kabsch of each patch match's correspondences -> ICP from that on (src, tgt) -> displacement rows [s, T s].

    For parity with the reference, (src, tgt) are the MUTUAL points of the match (pass ``corr_src, corr_off, corr_ref,
    corr_off``; :3352-3353) and ``rows_src / rows_off`` all points of the source patch (:3348, 3371-3374); without
    ``rows_src`` the rows are those of ``src``.  Matches with fewer than ``min_corr`` correspondences are skipped as the
    reference skips them (:3338): ``iters == -1``, identity transform, rows left unwritten (zero here).
    ``init_round_f32``: ICP starts from the float32 values of the Kabsch transform, like the reference's float32 4 x 4
    (scripts/weighted_svd.py:148-151, :3360).

    Equivalent to ``T0 = kabsch_transforms(...); out = piecewise_icp(..., init_T=T0); rows = apply_transform(rows_src,
    rows_off, out["T"])``.  Returns the dict of :func:`piecewise_icp` plus ``rows`` (n_rows, 6) float32 when ``return_rows``."""
    torch = require_gpu()
    if icp_type not in _ICP_MODES:
        raise ValueError("ICP type not supported")  # utils/o3d_tools.py:43
    mode = _ICP_MODES[icp_type] | _p2plane_bit(p2plane)
    src = _dev(src, torch.float32, "src", (3,))
    tgt = _dev(tgt, torch.float32, "tgt", (3,))
    src_off = _dev(src_off, torch.int64, "src_off")
    tgt_off = _dev(tgt_off, torch.int64, "tgt_off")
    corr_src = _dev(corr_src, torch.float32, "corr_src", (3,))
    corr_ref = _dev(corr_ref, torch.float32, "corr_ref", (3,))
    corr_off = _dev(corr_off, torch.int64, "corr_off")
    cw = None if corr_weights is None else _dev(corr_weights, torch.float32, "corr_weights")
    P = src_off.shape[0] - 1
    if tgt_off.shape[0] - 1 != P or corr_off.shape[0] - 1 != P:
        raise ValueError("src_off, tgt_off and corr_off must describe the same number of patches")
    if corr_ref.shape[0] != corr_src.shape[0] or (cw is not None and cw.shape[0] != corr_src.shape[0]):
        raise ValueError("corr_src, corr_ref and corr_weights must have the same number of rows")
    if max_src_patch is None:
        max_src_patch = _max_patch(src_off)
    if max_tgt_patch is None:
        max_tgt_patch = _max_patch(tgt_off)
    tn, nbit = None, 0
    if icp_type == "point2plane":
        tn, nbit = _target_normals(torch, tgt, tgt_off, max_tgt_patch, tgt_normals)
    dev = src.device
    T = torch.empty((P, 4, 4), dtype=torch.float64, device=dev)
    fit = torch.empty((P,), dtype=torch.float64, device=dev)
    rmse = torch.empty((P,), dtype=torch.float64, device=dev)
    iters = torch.empty((P,), dtype=torch.int32, device=dev)
    corr = torch.empty((src.shape[0],), dtype=torch.int32, device=dev) if return_corr else None
    if (rows_src is None) != (rows_off is None):
        raise ValueError("rows_src and rows_off go together")
    if rows_src is not None:
        rows_src = _dev(rows_src, torch.float32, "rows_src", (3,))
        rows_off = _dev(rows_off, torch.int64, "rows_off")
        if rows_off.shape[0] - 1 != P:
            raise ValueError("rows_off must describe the same number of patches")
    n_rows = src.shape[0] if rows_src is None else rows_src.shape[0]
    # (zero filled only when matches can be skipped: their rows stay unwritten)
    rows = (torch.zeros if min_corr > 0 else torch.empty)((n_rows, 6), dtype=torch.float32, device=dev) if return_rows else None
    check(lib().f4l_patch_loop(ptr(src), ptr(src_off), ptr(tgt), ptr(tgt_off), P, ptr(corr_src), ptr(corr_ref), ptr(cw),
                               ptr(corr_off), int(min_corr), float(weight_thresh), float(eps), ptr(tn), float(max_corr_dist),
                               int(max_iter), float(rel_fitness), float(rel_rmse), mode | nbit | (0x100 if init_round_f32 else 0),
                               int(bool(fixed_iters)),
                               {"f32": _lib.SEARCH_F32, "f64": _lib.SEARCH_F64}[search], int(max_src_patch),
                               int(max_tgt_patch), int(src.shape[0]), ptr(T), ptr(fit), ptr(rmse), ptr(iters), ptr(corr),
                               ptr(rows_src), ptr(rows_off), ptr(rows), stream_ptr()), "f4l_patch_loop")
    out = dict(T=T, fitness=fit, rmse=rmse, iters=iters)
    if return_corr:
        out["corr"] = corr
    if return_rows:
        out["rows"] = rows
    return out


def merge_tiles(tiles):
    """The CSR arrays of several tiles' patch matches concatenated (offsets shifted) into the arguments of ONE :func:`patch_loop`:
    dict(src, src_off, tgt, tgt_off, corr_src, corr_ref, corr_off[, corr_weights, rows_src, rows_off][, max_src_patch,
    max_tgt_patch]) plus `split`, the per-tile (patches, rows, source points) counts :func:`split_tiles` cuts the results by.
    tiles: list of dicts with those keys (all tiles with or all without the optional ones; `max_src` / `max_tgt`, the tile's largest
    patch sizes, spare the launch its read-back)."""
    torch = require_gpu()

    def cat_pts(key):
        return torch.cat([t[key] for t in tiles]) if tiles[0].get(key) is not None else None

    def cat_off(key, pts_key):
        if tiles[0].get(key) is None:
            return None
        parts, base = [], 0
        for i, t in enumerate(tiles):
            parts.append((t[key] if i == 0 else t[key][1:]) + base)
            base += t[pts_key].shape[0]  # (= the tile's last offset, known without asking the device)
        return torch.cat(parts)
    m = {k: cat_pts(k) for k in ("src", "tgt", "corr_src", "corr_ref", "corr_weights", "rows_src")}
    m.update({k: cat_off(k, v) for k, v in (("src_off", "src"), ("tgt_off", "tgt"), ("corr_off", "corr_src"), ("rows_off", "rows_src"))})
    # (the fine matching registers the mutual points themselves: its tiles pass the same arrays as cloud and as matches)
    for dup, of in (("corr_src", "src"), ("corr_ref", "tgt"), ("corr_off", "src_off"), ("tgt_off", "src_off")):
        if all(t.get(dup) is t.get(of) for t in tiles):
            m[dup] = m[of]
    for key, name in (("max_src", "max_src_patch"), ("max_tgt", "max_tgt_patch")):
        if all(key in t for t in tiles):
            m[name] = max(int(t[key]) for t in tiles)
    m["split"] = [(t["src_off"].shape[0] - 1, (t["rows_src"] if t.get("rows_src") is not None else t["src"]).shape[0], t["src"].shape[0]) for t in tiles]
    return m


def split_tiles(out, split):
    """The result of a merged launch cut back into one dict per tile (views)."""
    res, p0, r0, s0 = [], 0, 0, 0
    for P, n_rows, n_src in split:
        one = {k: out[k][p0:p0 + P] for k in ("T", "fitness", "rmse", "iters")}
        if "rows" in out:
            one["rows"] = out["rows"][r0:r0 + n_rows]
        if "corr" in out:
            one["corr"] = out["corr"][s0:s0 + n_src]
        res.append(one)
        p0, r0, s0 = p0 + P, r0 + n_rows, s0 + n_src
    return res


def patch_loop_tiles(tiles, **kw):
    """The loop body of SEVERAL tiles in one launch.  The reference works tile by tile (main_fusion.py:134: <= 1 M points each,
    configs/landslide/fusion_brienz.yaml:25-26), and one tile's ~2000 patch matches are only two rounds of workgroups on this
    chip -- a launch of that size runs at two thirds of the rate of a large one.  Patches are independent and
    f4l_patch_loop does not care which tile a patch came from: the tiles' CSR arrays are concatenated (:func:`merge_tiles`),
    launched once, and the per-patch results and rows handed back per tile as views (:func:`split_tiles`).  A pipeline that
    produces its tiles' arrays into one buffer in the first place calls :func:`patch_loop` on that directly.
    `kw`: the keyword arguments of :func:`patch_loop`.  Returns a list of result dicts, one per tile."""
    if not tiles:
        return []
    m = merge_tiles(tiles)
    split = m.pop("split")
    args = [m.pop(k) for k in ("src", "src_off", "tgt", "tgt_off", "corr_src", "corr_ref", "corr_off", "corr_weights")]
    for name in ("max_src_patch", "max_tgt_patch"):
        if name in kw:
            m.pop(name, None)
    return split_tiles(patch_loop(*args, **m, **kw), split)


def mutual_correspondences(src_ids, src_off, tgt_ids, tgt_off, corr_tgt):
    """`torch.isin(corr[src patch][:, 1], tgt patch)` for all patch matches at once (src/coarse_to_fine_matching_base.py:
    3259-3274; f4l_mutual_correspondences).  src_ids / tgt_ids: int64 point ids grouped by match (CSR offsets src_off /
    tgt_off), ids ascending inside every target patch; corr_tgt (n_src_points,) int64: the target point matched to each
    source point, -1 for none.  Returns (mask (len(src_ids),) bool, count (P,) int64)."""
    torch = require_gpu()
    src_ids = _dev(src_ids, torch.int64, "src_ids")
    tgt_ids = _dev(tgt_ids, torch.int64, "tgt_ids")
    src_off = _dev(src_off, torch.int64, "src_off")
    tgt_off = _dev(tgt_off, torch.int64, "tgt_off")
    corr_tgt = _dev(corr_tgt, torch.int64, "corr_tgt")
    P = src_off.shape[0] - 1
    if tgt_off.shape[0] - 1 != P:
        raise ValueError("src_off and tgt_off must describe the same number of patch matches")
    mask = torch.empty((src_ids.shape[0],), dtype=torch.uint8, device=src_ids.device)
    count = torch.empty((P,), dtype=torch.int64, device=src_ids.device)
    check(lib().f4l_mutual_correspondences(ptr(src_ids), ptr(src_off), ptr(tgt_ids), ptr(tgt_off), P, ptr(corr_tgt),
                                           corr_tgt.shape[0], ptr(mask), ptr(count), stream_ptr()), "f4l_mutual_correspondences")
    return mask.to(torch.bool), count


def rigidity_check(corr_src, corr_ref, corr_off, thres_dist_diff, precision="f64"):
    """The quality test of a patch match before the rigid fit (src/coarse_to_fine_matching_base.py:3304-3320,
    f4l_rigidity_check): per match, the mean of |d(s_i, s_j) - d(t_i, t_j)| over its mutual pairs and the share of pairs
    with that difference <= thres_dist_diff.  precision "f64" (parity mode) or "f32" (f4l_rigidity_check_f32: float32 pair
    arithmetic, distances within 3e-7 of themselves).  Returns (dist_mean (P,), ratio_inlier (P,)) float64."""
    if precision not in ("f64", "f32"):
        raise ValueError("precision must be 'f64' or 'f32', not %r" % (precision,))
    torch = require_gpu()
    corr_src = _dev(corr_src, torch.float32, "corr_src", (3,))
    corr_ref = _dev(corr_ref, torch.float32, "corr_ref", (3,))
    corr_off = _dev(corr_off, torch.int64, "corr_off")
    P = corr_off.shape[0] - 1
    dm = torch.empty((P,), dtype=torch.float64, device=corr_src.device)
    ri = torch.empty((P,), dtype=torch.float64, device=corr_src.device)
    name = "f4l_rigidity_check" if precision == "f64" else "f4l_rigidity_check_f32"
    check(getattr(lib(), name)(ptr(corr_src), ptr(corr_ref), ptr(corr_off), P, float(thres_dist_diff), ptr(dm), ptr(ri), stream_ptr()), name)
    return dm, ri


def apply_transform(pts, off, T, inverse=False):
    """Rows [s, T_p s] (src/coarse_to_fine_matching_base.py:3371-3374,3408) -> (n, 6) float32."""
    torch = require_gpu()
    pts = _dev(pts, torch.float32, "pts", (3,))
    off = _dev(off, torch.int64, "off")
    T = _dev(T, torch.float64, "T")
    n, P = pts.shape[0], off.shape[0] - 1
    out = torch.empty((n, 6), dtype=torch.float32, device=pts.device)
    check(lib().f4l_apply_transform(ptr(pts), ptr(off), P, n, ptr(T), int(bool(inverse)), ptr(out), stream_ptr()),
          "f4l_apply_transform")
    return out


def nn_refine(src, src_off, tgt, tgt_off, T, thr, max_tgt_patch=None, return_rows=True):
    """`refine_dvfs_with_threshold` for all patches (src/coarse_to_fine_matching_base.py:48-97).

    Returns (nn (n_src,) int32 index inside the target patch or -1, rows (n_src, 6) float32 or None)."""
    torch = require_gpu()
    src = _dev(src, torch.float32, "src", (3,))
    tgt = _dev(tgt, torch.float32, "tgt", (3,))
    src_off = _dev(src_off, torch.int64, "src_off")
    tgt_off = _dev(tgt_off, torch.int64, "tgt_off")
    T = _dev(T, torch.float64, "T")
    thr = _dev(thr, torch.float64, "thr")
    P = src_off.shape[0] - 1
    if max_tgt_patch is None:
        max_tgt_patch = _max_patch(tgt_off)
    nn = torch.empty((src.shape[0],), dtype=torch.int32, device=src.device)
    rows = torch.empty((src.shape[0], 6), dtype=torch.float32, device=src.device) if return_rows else None
    check(lib().f4l_nn_refine(ptr(src), ptr(src_off), ptr(tgt), ptr(tgt_off), P, ptr(T), ptr(thr),
                              int(max_tgt_patch), ptr(nn), ptr(rows), stream_ptr()), "f4l_nn_refine")
    return nn, rows


def match_lists(src, src_off, tgt, tgt_off, nn):
    """`nn_refine`'s answers (index inside the target patch, -1 for none) as the correspondence lists of :func:`patch_loop`:
    (corr_src (m, 3), corr_ref (m, 3), corr_off (P + 1,)) -- the rows that found a match, in row order (f4l_match_lists)."""
    torch = require_gpu()
    src = _dev(src, torch.float32, "src", (3,))
    tgt = _dev(tgt, torch.float32, "tgt", (3,))
    src_off = _dev(src_off, torch.int64, "src_off")
    tgt_off = _dev(tgt_off, torch.int64, "tgt_off")
    nn = _dev(nn, torch.int32, "nn")
    P, n = src_off.shape[0] - 1, src.shape[0]
    before = torch.zeros((n + 1,), dtype=torch.int64, device=src.device)
    torch.cumsum(nn >= 0, 0, out=before[1:])
    m = int(before[-1].item())
    cs = torch.empty((m, 3), dtype=torch.float32, device=src.device)
    ct = torch.empty((m, 3), dtype=torch.float32, device=src.device)
    coff = torch.empty((P + 1,), dtype=torch.int64, device=src.device)
    check(lib().f4l_match_lists(ptr(src), ptr(src_off), ptr(tgt), ptr(tgt_off), P, ptr(nn), ptr(before), ptr(cs), ptr(ct), ptr(coff),
                                stream_ptr()), "f4l_match_lists")
    return cs, ct, coff


def knn(xyz, k, return_d2=False):
    """Exact kNN of every point inside the cloud (kd_tree.h:266-280 semantics) -> (n, k) int32[, (n, k) f64]."""
    torch = require_gpu()
    xyz = _dev(xyz, torch.float32, "xyz", (3,))
    n = xyz.shape[0]
    idx = torch.empty((n, k), dtype=torch.int32, device=xyz.device)
    d2 = torch.empty((n, k), dtype=torch.float64, device=xyz.device) if return_d2 else None
    nbytes = lib().f4l_knn_workspace_bytes(n, k)
    ws = _scratch(nbytes, xyz.device)
    check(lib().f4l_knn(ptr(xyz), n, int(k), ptr(idx), ptr(d2), ptr(ws), C.c_size_t(nbytes), stream_ptr()), "f4l_knn")
    return (idx, d2) if return_d2 else idx


def knn_normals(xyz, k, return_d2=False, return_nn1=False):
    """Exact kNN and the PCA normal of every neighbour list in ONE launch (f4l_knn_normals; supervoxel.cpp:105-113)
    -> (n, k) int32, (n, 3) float64[, (n, k) f64][, (n,) f64: squared distance to the nearest other point
    (f4l_knn_normals_nn1; equal to `knn(xyz, 2, return_d2=True)[1][:, 1]`)]."""
    torch = require_gpu()
    xyz = _dev(xyz, torch.float32, "xyz", (3,))
    n = xyz.shape[0]
    idx = torch.empty((n, k), dtype=torch.int32, device=xyz.device)
    nrm = torch.empty((n, 3), dtype=torch.float64, device=xyz.device)
    d2 = torch.empty((n, k), dtype=torch.float64, device=xyz.device) if return_d2 else None
    nbytes = lib().f4l_knn_workspace_bytes(n, k)
    ws = _scratch(nbytes, xyz.device)
    if return_nn1:
        nn1 = torch.empty((n,), dtype=torch.float64, device=xyz.device)
        check(lib().f4l_knn_normals_nn1(ptr(xyz), n, int(k), ptr(idx), ptr(d2), ptr(nrm), ptr(nn1), ptr(ws), C.c_size_t(nbytes),
                                        stream_ptr()), "f4l_knn_normals_nn1")
        return (idx, nrm, d2, nn1) if return_d2 else (idx, nrm, nn1)
    check(lib().f4l_knn_normals(ptr(xyz), n, int(k), ptr(idx), ptr(d2), ptr(nrm), ptr(ws), C.c_size_t(nbytes), stream_ptr()),
          "f4l_knn_normals")
    return (idx, nrm, d2) if return_d2 else (idx, nrm)


def nn_query(cloud, queries, k=1, return_d2=False):
    """The k nearest points of `cloud` for every query point -- `cKDTree(cloud).query(queries, k)` as used by
    `_voxel_subsampling` (src/coarse_to_fine_matching_base.py:1042-1046) -> (m, k) int32[, (m, k) f64 squared]."""
    torch = require_gpu()
    cloud = _dev(cloud, torch.float32, "cloud", (3,))
    queries = _dev(queries, torch.float32, "queries", (3,))
    n, m = cloud.shape[0], queries.shape[0]
    idx = torch.empty((m, k), dtype=torch.int32, device=cloud.device)
    d2 = torch.empty((m, k), dtype=torch.float64, device=cloud.device) if return_d2 else None
    nbytes = lib().f4l_nn_query_workspace_bytes(n, m, int(k))
    ws = _scratch(nbytes, cloud.device)
    check(lib().f4l_nn_query(ptr(cloud), n, ptr(queries), m, int(k), ptr(idx), ptr(d2), ptr(ws), C.c_size_t(nbytes),
                             stream_ptr()), "f4l_nn_query")
    return (idx, d2) if return_d2 else idx


def voxel_downsample(xyz, voxel_size, return_map=False, layout="open3d"):
    """Open3D `voxel_down_sample(voxel_size)` (src/coarse_to_fine_matching_base.py:1024-1025), or with layout="pcl" the
    `pcl::VoxelGrid` of cpp_core/pcd_tiling/pcd_tiling.cpp:118-227: the mean point of every occupied voxel -> (M, 3)
    float64, voxels in ascending (z, y, x) index order[, points per voxel (M,) int32, voxel of every input point (n,)
    int32]."""
    torch = require_gpu()
    xyz = _dev(xyz, torch.float32, "xyz", (3,))
    n = xyz.shape[0]
    pts = torch.empty((n, 3), dtype=torch.float64, device=xyz.device)
    cnt = torch.empty((n,), dtype=torch.int32, device=xyz.device) if return_map else None
    vop = torch.empty((n,), dtype=torch.int32, device=xyz.device) if return_map else None
    m = C.c_int64(0)
    nbytes = lib().f4l_voxel_downsample_workspace_bytes(n)
    ws = _scratch(nbytes, xyz.device)
    check(lib().f4l_voxel_downsample(ptr(xyz), n, float(voxel_size), {"open3d": 0, "pcl": 1}[layout], ptr(pts), ptr(cnt),
                                     ptr(vop), C.byref(m), ptr(ws),
                                     C.c_size_t(nbytes), stream_ptr()), "f4l_voxel_downsample")
    M = int(m.value)
    return (pts[:M], cnt[:M], vop) if return_map else pts[:M]


def voxel_subsampling(src, tgt=None, voxel_size=None):
    """`_voxel_subsampling` (src/coarse_to_fine_matching_base.py:1012-1057) for one cloud or both epochs: the adaptive
    voxel size (`_compute_median_resolution`, :1022) unless given, the voxel-grid filter, the index of the original point
    nearest to every voxel centre (`idx_voxel2pts`, :1042-1049) and its inverse (`idx_pts2voxel`, -1 for points that
    represent no voxel, :1052-1059).  Returns a dict (per cloud: `pts_sub` float32 like pcd2tensor, `idx_voxel2pts`
    int64, `idx_pts2voxel` int64) plus `voxel_size`."""
    torch = require_gpu()
    src = _dev(src, torch.float32, "src", (3,))
    tgt = None if tgt is None else _dev(tgt, torch.float32, "tgt", (3,))
    if voxel_size is None:
        voxel_size = median_resolution(src, tgt)

    def one(xyz):
        sub = voxel_downsample(xyz, voxel_size).to(torch.float32)  # pcd2tensor casts to float32 (utils/o3d_tools.py:241-257)
        v2p = nn_query(xyz, sub, 1)[:, 0].to(torch.int64)
        p2v = torch.full((xyz.shape[0],), -1, dtype=torch.int64, device=xyz.device)
        # (a point that is nearest to two voxel centres keeps the larger voxel index: what the reference's in-order
        #  CPU assignment gives; its CUDA assignment leaves that case undefined)
        p2v.scatter_reduce_(0, v2p, torch.arange(v2p.shape[0], dtype=torch.int64, device=xyz.device), "amax")
        return dict(pts_sub=sub, idx_voxel2pts=v2p, idx_pts2voxel=p2v)

    out = dict(voxel_size=float(voxel_size), src=one(src))
    if tgt is not None:
        out["tgt"] = one(tgt)
    return out


def normals(xyz, knn_idx):
    """PCA normals from neighbour lists (pca_estimate_normals.h:43-108) -> (n, 3) float64."""
    torch = require_gpu()
    xyz = _dev(xyz, torch.float32, "xyz", (3,))
    knn_idx = _dev(knn_idx, torch.int32, "knn_idx")
    n, k = knn_idx.shape
    out = torch.empty((n, 3), dtype=torch.float64, device=xyz.device)
    check(lib().f4l_normals(ptr(xyz), n, ptr(knn_idx), int(k), ptr(out), stream_ptr()), "f4l_normals")
    return out


def supervoxel(xyz, k, resolution, return_intermediates=False):
    """Whole partition (supervoxel.cpp:92-133 without file I/O). Returns labels (n,) int32 on the GPU and K.

    The reference's own labels: kNN, normals AND the order-dependent segmentation run on the GPU (csrc/supervoxel_exact.hip: the
    sequential fusion and the FIFO exchange as fixed points of parallel passes; the one-core host replay is the fall-back and
    F4L_SV_EXACT_HOST=1).  The call synchronises the current stream (it looks at the convergence of the passes)."""
    torch = require_gpu()
    xyz = _dev(xyz, torch.float32, "xyz", (3,))
    n = xyz.shape[0]
    labels = torch.empty((n,), dtype=torch.int32, device=xyz.device)
    knn_out = torch.empty((n, k), dtype=torch.int32, device=xyz.device) if return_intermediates else None
    nrm_out = torch.empty((n, 3), dtype=torch.float64, device=xyz.device) if return_intermediates else None
    nbytes = lib().f4l_supervoxel_workspace_bytes(n, k)
    ws = torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=xyz.device)
    nsv = C.c_int32(0)
    check(lib().f4l_supervoxel(ptr(xyz), n, int(k), float(resolution), ptr(labels), C.byref(nsv), ptr(knn_out),
                               ptr(nrm_out), ptr(ws), C.c_size_t(nbytes), stream_ptr()), "f4l_supervoxel")
    if return_intermediates:
        return labels, nsv.value, knn_out, nrm_out
    return labels, nsv.value


def supervoxel_segment_device(xyz, normals, knn_idx, resolution, return_reps=False, grid_bbox=None):
    """The segmentation stage entirely on the device, asynchronously (f4l_supervoxel_segment_device: the parallel variant
    of supervoxel_segmentation.h:65-248; NOT label-identical to the sequential reference, same invariants).
    Returns labels (n,) int32 and info (8,) int32 = [supervoxels, K wanted, status bits, sweeps, the starting lambda's two
    32-bit words (`supervoxel_lambda0(info)`), lambda rounds, 1 + the sub-round that was cut to reach K (0: none)], both ON THE
    DEVICE (reading `info` is the caller's synchronisation point)[, reps (n,) int32: the first info[0] entries are the
    representative point of every supervoxel].  `grid_bbox` (6 floats: min xyz, max xyz) anchors the resolution grid whose
    occupied cells set the count (default: the cloud's own box); knn entries < 0 or equal to their row are "no neighbour"."""
    torch = require_gpu()
    xyz = _dev(xyz, torch.float32, "xyz", (3,))
    normals = _dev(normals, torch.float64, "normals", (3,))
    knn_idx = _dev(knn_idx, torch.int32, "knn_idx")
    n, k = knn_idx.shape
    if xyz.shape[0] != n or normals.shape[0] != n:
        raise ValueError("xyz, normals and knn_idx must describe the same points")
    labels = torch.empty((n,), dtype=torch.int32, device=xyz.device)
    info = torch.zeros((8,), dtype=torch.int32, device=xyz.device)
    reps = torch.empty((n,), dtype=torch.int32, device=xyz.device) if return_reps else None
    nbytes = lib().f4l_supervoxel_segment_device_workspace_bytes(n, k)
    ws = torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=xyz.device)
    box = None if grid_bbox is None else (C.c_float * 6)(*[float(v) for v in grid_bbox])
    check(lib().f4l_supervoxel_segment_device(ptr(xyz), ptr(normals), ptr(knn_idx), n, int(k), float(resolution), box, ptr(labels),
                                              ptr(reps), ptr(info), ptr(ws), C.c_size_t(nbytes), stream_ptr()),
          "f4l_supervoxel_segment_device")
    return (labels, info, reps) if return_reps else (labels, info)


def supervoxel_lambda0(info):
    """The fusion's starting lambda (supervoxel_segmentation.h:105-113) out of the `info` words of the device segmentation."""
    import struct
    w = [int(v) & 0xffffffff for v in info[4:6]]
    return struct.unpack("<d", struct.pack("<II", w[0], w[1]))[0]


def supervoxel_parallel(xyz, k, resolution, return_intermediates=False, read_count=True):
    """Whole partition on the device (f4l_supervoxel_parallel: kNN + normals + the parallel segmentation).  Returns labels
    (n,) int32 on the GPU and K (reads the device-side count: the one synchronisation)[, knn, normals, reps, info].
    read_count=False: nothing is read back -- returns (labels, info (8,) int32 ON THE DEVICE[, knn, normals, reps]); with the
    stream under capture (or F4L_KNN_ASYNC set) f4l_knn sizes its grid on the device too, and the whole partition is one
    enqueue-only call that a HIP graph can hold (tests/test_gpu_supervoxel_parallel.py)."""
    torch = require_gpu()
    xyz = _dev(xyz, torch.float32, "xyz", (3,))
    n = xyz.shape[0]
    labels = torch.empty((n,), dtype=torch.int32, device=xyz.device)
    info = torch.zeros((8,), dtype=torch.int32, device=xyz.device)
    knn_out = torch.empty((n, k), dtype=torch.int32, device=xyz.device) if return_intermediates else None
    nrm_out = torch.empty((n, 3), dtype=torch.float64, device=xyz.device) if return_intermediates else None
    reps = torch.empty((n,), dtype=torch.int32, device=xyz.device) if return_intermediates else None
    nbytes = lib().f4l_supervoxel_parallel_workspace_bytes(n, k)
    ws = torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=xyz.device)
    check(lib().f4l_supervoxel_parallel(ptr(xyz), n, int(k), float(resolution), ptr(labels), ptr(reps), ptr(info), ptr(knn_out),
                                        ptr(nrm_out), ptr(ws), C.c_size_t(nbytes), stream_ptr()), "f4l_supervoxel_parallel")
    if not read_count:
        return (labels, info, knn_out, nrm_out, reps) if return_intermediates else (labels, info)
    info_h = info.cpu()
    K = int(info_h[0])
    if int(info_h[2]) & 14:  # (bit 0, a disconnected neighbour graph that stops above its target, leaves a valid partition)
        raise RuntimeError(f"f4l_supervoxel_parallel: the segmentation did not finish (status bits {int(info_h[2])}: 2 = lambda "
                           "schedule exhausted above K, 4 = exchange stopped by its sweep budget)")
    if return_intermediates:
        return labels, K, knn_out, nrm_out, reps[:K], info_h
    return labels, K


class PartitionNeighbours:
    """What f4l_partition_neighbours left for f4l_partition_segment: the workspace (neighbour lists, normals and the cloud in the
    search's own order, inside it) and, when asked for, every point's squared distance to its nearest other point."""
    def __init__(self, ws, nbytes, n, k, nn1_d2):
        self.ws, self.nbytes, self.n, self.k, self.nn1_d2 = ws, nbytes, n, k, nn1_d2


def partition_neighbours(xyz, k, return_nn1=False):
    """First half of the partition (f4l_partition_neighbours): the neighbour search and the normals, which do not depend on the
    resolution -- `_compute_median_resolution` (src/coarse_to_fine_matching_base.py:2716-2754) can take its nearest-neighbour
    distances from `.nn1_d2` (n,) float64, in the search's order, before `partition_segment` is given the resolution."""
    torch = require_gpu()
    xyz = _dev(xyz, torch.float32, "xyz", (3,))
    n = xyz.shape[0]
    nbytes = lib().f4l_partition_workspace_bytes(n, int(k))
    ws = torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=xyz.device)
    nn1 = torch.empty((n,), dtype=torch.float64, device=xyz.device) if return_nn1 else None
    check(lib().f4l_partition_neighbours(ptr(xyz), n, int(k), ptr(nn1), ptr(ws), C.c_size_t(nbytes), stream_ptr()), "f4l_partition_neighbours")
    return PartitionNeighbours(ws, nbytes, n, int(k), nn1)


def partition_segment(nb, resolution, return_reps=False, grid_bbox=None):
    """Second half (f4l_partition_segment): labels (n,) int32 in the caller's order and info (8,) int32 on the device (see
    supervoxel_segment_device)[, reps]."""
    torch = require_gpu()
    dev = nb.ws.device
    labels = torch.empty((nb.n,), dtype=torch.int32, device=dev)
    info = torch.zeros((8,), dtype=torch.int32, device=dev)
    reps = torch.empty((nb.n,), dtype=torch.int32, device=dev) if return_reps else None
    box = None
    if grid_bbox is not None:
        box = (C.c_float * 6)(*[float(v) for v in grid_bbox])
    check(lib().f4l_partition_segment(nb.n, nb.k, float(resolution), box, ptr(labels), ptr(reps), ptr(info), ptr(nb.ws),
                                      C.c_size_t(nb.nbytes), stream_ptr()), "f4l_partition_segment")
    return (labels, info, reps) if return_reps else (labels, info)


def labels_to_csr(labels, K):
    """Sort-by-label -> (order (n,) int32, off (K+1,) int64); replaces prepare_pts2spt_dict's mask loop
    (src/coarse_to_fine_matching_base.py:1327-1332)."""
    torch = require_gpu()
    labels = _dev(labels, torch.int32, "labels")
    n = labels.shape[0]
    order = torch.empty((n,), dtype=torch.int32, device=labels.device)
    off = torch.empty((K + 1,), dtype=torch.int64, device=labels.device)
    nbytes = lib().f4l_labels_to_csr_workspace_bytes(n, K)
    ws = _scratch(nbytes, labels.device)
    check(lib().f4l_labels_to_csr(ptr(labels), n, int(K), ptr(order), ptr(off), ptr(ws), C.c_size_t(nbytes),
                                  stream_ptr()), "f4l_labels_to_csr")
    return order, off


def labels_to_csr_via(labels, via, K):
    """`labels_to_csr(labels[via], K)` without the gathered labels: row i of `via` (int32 rows of `labels`; outside = no patch)
    belongs to patch labels[via[i]] -> (order (m,) int32, off (K+1,) int64).  f4l_labels_to_csr_via."""
    torch = require_gpu()
    labels = _dev(labels, torch.int32, "labels")
    via = _dev(via, torch.int32, "via")
    m = via.shape[0]
    order = torch.empty((m,), dtype=torch.int32, device=labels.device)
    off = torch.empty((K + 1,), dtype=torch.int64, device=labels.device)
    nbytes = lib().f4l_labels_to_csr_workspace_bytes(m, K)
    ws = _scratch(nbytes, labels.device)
    check(lib().f4l_labels_to_csr_via(ptr(labels), labels.shape[0], ptr(via), m, int(K), ptr(order), ptr(off), ptr(ws),
                                      C.c_size_t(nbytes), stream_ptr()), "f4l_labels_to_csr_via")
    return order, off


def epoch_join(src, tgt, return_nn1=True):
    """The two searches over the second epoch with ONE binning of it (f4l_epoch_join): -> (tgt_to_src (m,) int32 = every target
    point's nearest source point, the `nn_query(src, tgt, 1)` of the label transfer; tgt_nn1_d2 (m,) float64 = its squared
    distance to the nearest other target point, what `_compute_median_resolution` (src/coarse_to_fine_matching_base.py:2716-2754)
    takes the median of -- or None)."""
    torch = require_gpu()
    src = _dev(src, torch.float32, "src", (3,))
    tgt = _dev(tgt, torch.float32, "tgt", (3,))
    n, m = src.shape[0], tgt.shape[0]
    idx = torch.empty((m,), dtype=torch.int32, device=src.device)
    nn1 = torch.empty((m,), dtype=torch.float64, device=src.device) if return_nn1 else None
    nbytes = lib().f4l_epoch_join_workspace_bytes(n, m)
    ws = _scratch(nbytes, src.device)
    check(lib().f4l_epoch_join(ptr(src), n, ptr(tgt), m, ptr(nn1), ptr(idx), ptr(ws), C.c_size_t(nbytes), stream_ptr()), "f4l_epoch_join")
    return idx, nn1


def gather_points(pts, order):
    torch = require_gpu()
    pts = _dev(pts, torch.float32, "pts", (3,))
    order = _dev(order, torch.int32, "order")
    out = torch.empty((order.shape[0], 3), dtype=torch.float32, device=pts.device)
    check(lib().f4l_gather_points(ptr(pts), ptr(order), order.shape[0], ptr(out), stream_ptr()), "f4l_gather_points")
    return out
