"""Minimal PLY reader/writer for the vertex element (x, y, z[, colours ...]); ascii, binary little- and big-endian.
Replaces the two file boundaries of the hot path: `pcl::io::loadPLYFile` (supervoxel.cpp:89-90) and
`o3d.io.read_point_cloud` (src/piecewise_icp.py:80,87).  File I/O only -- nothing here is on the timed path."""
import numpy as np

_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
          "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
          "double": "f8", "float64": "f8"}


def read_ply(path):
    """Returns (xyz (n,3) float64, fields dict name -> (n,) array of the remaining vertex properties)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, n, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated header")
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("list properties on the vertex element are not supported")
                props.append((tok[2], _TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=n, ndmin=2) if n else np.zeros((0, len(props)))
            cols = {name: data[:, i] for i, (name, _) in enumerate(props)}
        else:
            end = "<" if fmt == "binary_little_endian" else ">"
            dt = np.dtype([(name, end + t) for name, t in props])
            rec = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
            cols = {name: rec[name] for name, _ in props}
    xyz = np.stack([np.asarray(cols[k], dtype=np.float64) for k in ("x", "y", "z")], axis=1) if n else np.zeros((0, 3))
    return xyz, {k: v for k, v in cols.items() if k not in ("x", "y", "z")}


def read_xyz32(path):
    """The vertex coordinates as a contiguous float32 (n, 3) array -- what `pcl::PointXYZ` (supervoxel.cpp:88-91) and the device
    keep of a tile.  Binary files whose x, y, z are float32 are sliced out of the record buffer without the float64 detour of
    `read_ply` (a third of its time on a 1 M-point tile); everything else goes through it."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, n, props, in_vertex, first = None, 0, [], False, None
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated header")
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if first is None:
                    first = tok[1]
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    props = None
                    break
                props.append((tok[2], _TYPES.get(tok[1])))
            elif tok[0] == "end_header":
                break
        fast = (props and fmt in ("binary_little_endian", "binary_big_endian") and first == "vertex"
                and all(t is not None for _, t in props) and all(dict(props).get(k) == "f4" for k in ("x", "y", "z")))
        if fast:
            end = "<" if fmt == "binary_little_endian" else ">"
            dt = np.dtype([(name, end + t) for name, t in props])
            rec = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
            out = np.empty((n, 3), dtype=np.float32)
            for d, k in enumerate(("x", "y", "z")):
                out[:, d] = rec[k]
            return out
    return np.ascontiguousarray(read_ply(path)[0], dtype=np.float32)


def write_ply(path, xyz, dtype="float32"):
    xyz = np.ascontiguousarray(xyz, dtype=dtype)
    t = "float" if dtype == "float32" else "double"
    with open(path, "wb") as f:
        f.write((f"ply\nformat binary_little_endian 1.0\nelement vertex {len(xyz)}\n"
                 f"property {t} x\nproperty {t} y\nproperty {t} z\nend_header\n").encode())
        f.write(xyz.astype("<" + ("f4" if dtype == "float32" else "f8")).tobytes())
