"""ctypes binding of libf4l_hip.so (the C ABI declared in include/f4l.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C fusion4landslide_amd/csrc``.
Loading is lazy; a missing library or a missing GPU is a hard error -- there is no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("F4L_LIB_PATH") or os.path.join(_HERE, "lib", "libf4l_hip.so")  # (override: A/B builds of the kernels)

F4L_OK = 0
ICP_POINT2POINT = 0
ICP_POINT2PLANE = 1
ICP_P2PL_OPEN3D = 0x400  # point-to-plane steps with Open3D's own semantics (include/f4l.h)
SEARCH_F32 = 0
SEARCH_F64 = 1
MAX_K = 64

# every symbol include/f4l.h declares, with its ctypes signature (all pointers as void*)
_P, _I, _I64, _D, _SZ = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_size_t
SIGNATURES = {
    "f4l_version": (C.c_int, []),
    "f4l_strerror": (C.c_char_p, [_I]),
    "f4l_last_hip_error": (C.c_int, []),
    "f4l_device_info": (C.c_int, [_P, _P, _P, _P, _I]),
    "f4l_kabsch_batched": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _D, _D, _P, _P, _P]),
    "f4l_kabsch_batched_f64": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _D, _D, _P, _P, _P]),
    "f4l_kabsch_transforms": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _D, _D, _P, _P]),
    "f4l_kabsch2_batched": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _I, _D, _D, _P, _P, _P]),
    "f4l_kabsch2_batched_f64": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _I, _D, _D, _P, _P, _P]),
    "f4l_kabsch_residuals": (C.c_int, [_P, _P, _P, _I64, _I64, _P, _P, _P, _P]),
    "f4l_piecewise_icp": (C.c_int, [_P, _P, _P, _P, _I64, _P, _P, _D, _I, _D, _D, _I, _I, _I, _I64, _I64, _I64, _P, _P,
                                    _P, _P, _P, _P]),
    "f4l_piecewise_gicp": (C.c_int, [_P, _P, _P, _P, _I64, _P, _P, _P, _D, _D, _I, _D, _D, _I, _I64, _I64, _I64, _P, _P, _P, _P, _P,
                                     _P]),
    "f4l_patch_loop": (C.c_int, [_P, _P, _P, _P, _I64, _P, _P, _P, _P, _I64, _D, _D, _P, _D, _I, _D, _D, _I, _I, _I, _I64, _I64, _I64, _P,
                                 _P, _P, _P, _P, _P, _P, _P, _P]),
    "f4l_mutual_correspondences": (C.c_int, [_P, _P, _P, _P, _I64, _P, _I64, _P, _P, _P]),
    "f4l_rigidity_check": (C.c_int, [_P, _P, _P, _I64, _D, _P, _P, _P]),
    "f4l_rigidity_check_f32": (C.c_int, [_P, _P, _P, _I64, _D, _P, _P, _P]),
    "f4l_patch_normals": (C.c_int, [_P, _P, _I64, _I, _I64, _P, _P]),
    "f4l_patch_normals_f64": (C.c_int, [_P, _P, _I64, _I, _I64, _P, _P]),
    "f4l_apply_transform": (C.c_int, [_P, _P, _I64, _I64, _P, _I, _P, _P]),
    "f4l_nn_refine": (C.c_int, [_P, _P, _P, _P, _I64, _P, _P, _I64, _P, _P, _P]),
    "f4l_knn_workspace_bytes": (_SZ, [_I64, _I]),
    "f4l_knn": (C.c_int, [_P, _I64, _I, _P, _P, _P, _SZ, _P]),
    "f4l_knn_normals": (C.c_int, [_P, _I64, _I, _P, _P, _P, _P, _SZ, _P]),
    "f4l_knn_normals_nn1": (C.c_int, [_P, _I64, _I, _P, _P, _P, _P, _P, _SZ, _P]),
    "f4l_normals": (C.c_int, [_P, _I64, _P, _I, _P, _P]),
    "f4l_voxel_downsample_workspace_bytes": (_SZ, [_I64]),
    "f4l_voxel_downsample": (C.c_int, [_P, _I64, _D, _I, _P, _P, _P, _P, _P, _SZ, _P]),
    "f4l_nn_query_workspace_bytes": (_SZ, [_I64, _I64, _I]),
    "f4l_nn_query": (C.c_int, [_P, _I64, _P, _I64, _I, _P, _P, _P, _SZ, _P]),
    "f4l_supervoxel_workspace_bytes": (_SZ, [_I64, _I]),
    "f4l_supervoxel": (C.c_int, [_P, _I64, _I, _D, _P, _P, _P, _P, _P, _SZ, _P]),
    "f4l_supervoxel_segment_device_workspace_bytes": (_SZ, [_I64, _I]),
    "f4l_supervoxel_segment_device": (C.c_int, [_P, _P, _P, _I64, _I, _D, _P, _P, _P, _P, _P, _SZ, _P]),
    "f4l_supervoxel_segment_exact_workspace_bytes": (_SZ, [_I64, _I]),
    "f4l_supervoxel_segment_exact": (C.c_int, [_P, _P, _P, _I64, _I, _D, _P, _P, _P, _P, _SZ, _P]),
    "f4l_partition_workspace_bytes": (_SZ, [_I64, _I]),
    "f4l_partition_neighbours": (C.c_int, [_P, _I64, _I, _P, _P, _SZ, _P]),
    "f4l_partition_segment": (C.c_int, [_I64, _I, _D, _P, _P, _P, _P, _P, _SZ, _P]),
    "f4l_supervoxel_parallel_workspace_bytes": (_SZ, [_I64, _I]),
    "f4l_supervoxel_parallel": (C.c_int, [_P, _I64, _I, _D, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "f4l_supervoxel_segment_host": (C.c_int, [_P, _P, _P, _I64, _I, _D, _P, _P]),
    "f4l_write_partition_txt": (C.c_int, [C.c_char_p, _P, _P, _I64, C.c_int32]),
    "f4l_write_rows_txt": (C.c_int, [C.c_char_p, _P, _I64, C.c_int]),
    "f4l_median_f64_workspace_bytes": (_SZ, [_I64]),
    "f4l_median_f64": (C.c_int, [_P, _I64, _I64, _P, _P, _SZ, _P]),
    "f4l_median_sqrt_f64": (C.c_int, [_P, _I64, _I64, _P, _P, _SZ, _P]),
    "f4l_labels_to_csr_workspace_bytes": (_SZ, [_I64, _I64]),
    "f4l_labels_to_csr": (C.c_int, [_P, _I64, _I64, _P, _P, _P, _SZ, _P]),
    "f4l_labels_to_csr_via": (C.c_int, [_P, _I64, _P, _I64, _I64, _P, _P, _P, _SZ, _P]),
    "f4l_epoch_join_workspace_bytes": (_SZ, [_I64, _I64]),
    "f4l_epoch_join": (C.c_int, [_P, _I64, _P, _I64, _P, _P, _P, _SZ, _P]),
    "f4l_match_lists": (C.c_int, [_P, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _P]),
    "f4l_gather_points": (C.c_int, [_P, _P, _I64, _P, _P]),
    "f4l_tile_point_clouds": (C.c_int, [C.c_char_p, C.c_char_p, _I, _I, _I, C.c_float, C.c_float, _I, C.c_char_p, _I, _P, _P]),
    "f4l_resave_point_cloud": (C.c_int, [C.c_char_p, C.c_char_p, _I, _P]),
}

_lib = None


class F4LError(RuntimeError):
    pass


def lib():
    """The loaded C-ABI library.  Raises F4LError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise F4LError(
                f"{LIB_PATH} is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C fusion4landslide_amd/csrc`. There is no CPU fallback for this package.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header/library drift
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != F4L_OK:
        L = lib()
        msg = L.f4l_strerror(rc).decode()
        extra = f" (hipError {L.f4l_last_hip_error()})" if rc == -3 else ""
        raise F4LError(f"{what}: {msg}{extra}")


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise F4LError("fusion4landslide_amd needs an AMD GPU (torch.cuda.is_available() is False); "
                       "there is no CPU fallback.")
    return torch


_EMPTY = {}  # device index -> a few bytes of device memory that stand for "an array of no elements"


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL).  An EMPTY tensor is still an array: torch hands out a null data pointer for
    it, which the C ABI reads as "argument missing" (a tile in which no patch has a single correspondence would be refused), so
    an empty CUDA tensor gets the address of a small buffer nobody reads."""
    if t is None:
        return None
    if t.numel() == 0 and t.is_cuda:
        buf = _EMPTY.get(t.device.index)
        if buf is None:
            import torch
            buf = _EMPTY[t.device.index] = torch.zeros(64, dtype=torch.uint8, device=t.device)
        return C.c_void_p(buf.data_ptr())
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
