// tiling.hip -- the tiling front end of the entry scripts as a library entry.
//
// cpp_core/pcd_tiling/pcd_tiling.cpp:709-871 `tile_point_clouds` (exported through pcd_tiling.h:3-17 and the SWIG module
// cpp_core/pcd_tiling/pcd_tiling.i): crop both epochs to the overlap of their bounding boxes (:73-116), thin them with a voxel grid
// (pcl::VoxelGrid, :118-227; leaf = median nearest-neighbour spacing of the smaller cloud when the size is 0, :37-54), halve the box
// along the longer in-plane side until both halves hold fewer than maxPointsPerTile points (:231-655) and write every leaf as
// non_overlap/{source,target}_tile_<i>.ply and overlap/{source,target}_tile_<i>_overlap.ply, the latter cut from the cloud with the
// leaf's box grown by 20 m in the projection plane.
//
// Rounds 2-5 did the crop, the recursion and the PLY output in numpy on the host around one device call (the voxel grid).  Here the
// clouds go to the device once and stay: bounding boxes, crops (stable compaction: rocPRIM select over a box predicate), the voxel
// grid with its colour averages, the spacing estimate (f4l_knn + a device rank selection) and the counts that steer the recursion
// are kernels; the host walks the tree of boxes (two counts per node decide it) and writes the leaves' PLY files.  The boxes of the
// recursion NEST (a child's box is its parent's, halved; a child's padded box lies inside its parent's padded box), so a leaf's
// clouds are the root clouds cropped to the leaf's own boxes: no intermediate cloud is materialised, and a node costs one counting
// pass instead of eight crops.
//
// PCL is not installable in the build container: the voxel filter and the crop follow PCL's documented behaviour [3P-knowledge,
// parity unpinned]; the output equals the numpy restatement of rounds 2-5 (oracle/pcd_tiling_ref.py) file for file, byte for byte.
// A file-level entry like f4l_write_partition_txt: it allocates its own device memory and synchronises.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <sys/stat.h>

#include <rocprim/rocprim.hpp>

#include "f4l_device.h"
#include "select.h"

namespace f4l {
namespace tiling {

constexpr double EPS = 1e-9;   // pcd_tiling.cpp:26
constexpr double PAD = 20.0;   // metres around a tile for its "overlap" twin (pcd_tiling.cpp:297-301 and every sibling branch)

// ---- PLY on the host ---------------------------------------------------------------------------------------------------------
struct HostCloud {
    std::vector<float> xyz;      // [n][3]: what pcl::PointXYZRGB keeps of a vertex
    std::vector<uint8_t> rgb;    // [n][3] or empty
    int64_t n = 0;
    bool has_rgb = false;
};
struct Prop { std::string name; int type; int size; };  // type: 0 i8 1 u8 2 i16 3 u16 4 i32 5 u32 6 f32 7 f64
static int prop_type(const std::string &t) {
    static const char *names[][2] = {{"char", "0"}, {"int8", "0"}, {"uchar", "1"}, {"uint8", "1"}, {"short", "2"}, {"int16", "2"}, {"ushort", "3"},
                                     {"uint16", "3"}, {"int", "4"}, {"int32", "4"}, {"uint", "5"}, {"uint32", "5"}, {"float", "6"}, {"float32", "6"},
                                     {"double", "7"}, {"float64", "7"}};
    for (auto &p : names)
        if (t == p[0]) return p[1][0] - '0';
    return -1;
}
static const int TYPE_SIZE[8] = {1, 1, 2, 2, 4, 4, 4, 8};
static double load_value(const unsigned char *p, int type, bool swap) {
    unsigned char b[8];
    const int sz = TYPE_SIZE[type];
    for (int i = 0; i < sz; ++i) b[i] = swap ? p[sz - 1 - i] : p[i];
    switch (type) {
        case 0: { int8_t v; memcpy(&v, b, 1); return v; }
        case 1: { uint8_t v; memcpy(&v, b, 1); return v; }
        case 2: { int16_t v; memcpy(&v, b, 2); return v; }
        case 3: { uint16_t v; memcpy(&v, b, 2); return v; }
        case 4: { int32_t v; memcpy(&v, b, 4); return v; }
        case 5: { uint32_t v; memcpy(&v, b, 4); return v; }
        case 6: { float v; memcpy(&v, b, 4); return v; }
        default: { double v; memcpy(&v, b, 8); return v; }
    }
}
// vertex element only: x, y, z (any scalar type, kept as float32) and red / green / blue (or r g b, or diffuse_*), ascii or binary
static int read_ply(const char *path, HostCloud &c) {
    FILE *fp = fopen(path, "rb");
    if (!fp) return F4L_EINVAL;
    std::vector<Prop> props;
    std::string fmt;
    int64_t n = 0;
    bool in_vertex = false, vertex_first = true, seen_element = false, ok = false;
    char line[1024];
    if (!fgets(line, sizeof line, fp) || strncmp(line, "ply", 3) != 0) { fclose(fp); return F4L_EINVAL; }
    while (fgets(line, sizeof line, fp)) {
        char a[256] = "", b[256] = "", d[256] = "";
        const int k = sscanf(line, "%255s %255s %255s", a, b, d);
        if (k < 1) continue;
        if (!strcmp(a, "format")) fmt = b;
        else if (!strcmp(a, "element")) {
            in_vertex = !strcmp(b, "vertex");
            if (in_vertex) { n = atoll(d); vertex_first = !seen_element; }
            seen_element = true;
        } else if (!strcmp(a, "property") && in_vertex) {
            if (!strcmp(b, "list")) { fclose(fp); return F4L_EUNSUPPORTED; }
            const int t = prop_type(b);
            if (t < 0) { fclose(fp); return F4L_EINVAL; }
            props.push_back({d, t, TYPE_SIZE[t]});
        } else if (!strcmp(a, "end_header")) { ok = true; break; }
    }
    if (!ok || n < 0 || !vertex_first) { fclose(fp); return F4L_EINVAL; }
    int ix = -1, iy = -1, iz = -1, ir = -1, ig = -1, ib = -1;
    for (int i = 0; i < (int)props.size(); ++i) {
        const std::string &s = props[(size_t)i].name;
        if (s == "x") ix = i; else if (s == "y") iy = i; else if (s == "z") iz = i;
    }
    static const char *rgb_names[3][3] = {{"red", "green", "blue"}, {"r", "g", "b"}, {"diffuse_red", "diffuse_green", "diffuse_blue"}};
    for (auto &names : rgb_names) {
        int f[3] = {-1, -1, -1};
        for (int i = 0; i < (int)props.size(); ++i)
            for (int ch = 0; ch < 3; ++ch)
                if (props[(size_t)i].name == names[ch]) f[ch] = i;
        if (f[0] >= 0 && f[1] >= 0 && f[2] >= 0) { ir = f[0]; ig = f[1]; ib = f[2]; break; }
    }
    if (ix < 0 || iy < 0 || iz < 0) { fclose(fp); return F4L_EINVAL; }
    c.n = n;
    c.has_rgb = ir >= 0;
    try {
        c.xyz.resize((size_t)n * 3);
        if (c.has_rgb) c.rgb.resize((size_t)n * 3);
    } catch (const std::bad_alloc &) { fclose(fp); return F4L_ENOMEM; }
    int rc = F4L_OK;
    if (fmt == "ascii") {
        std::vector<double> row(props.size());
        for (int64_t i = 0; i < n && rc == F4L_OK; ++i) {
            for (size_t p = 0; p < props.size(); ++p)
                if (fscanf(fp, "%lf", &row[p]) != 1) { rc = F4L_EINVAL; break; }
            if (rc != F4L_OK) break;
            c.xyz[3 * (size_t)i] = (float)row[(size_t)ix]; c.xyz[3 * (size_t)i + 1] = (float)row[(size_t)iy]; c.xyz[3 * (size_t)i + 2] = (float)row[(size_t)iz];
            if (c.has_rgb) { c.rgb[3 * (size_t)i] = (uint8_t)row[(size_t)ir]; c.rgb[3 * (size_t)i + 1] = (uint8_t)row[(size_t)ig]; c.rgb[3 * (size_t)i + 2] = (uint8_t)row[(size_t)ib]; }
        }
    } else if (fmt == "binary_little_endian" || fmt == "binary_big_endian") {
        const bool swap = fmt == "binary_big_endian";
        size_t stride = 0;
        std::vector<size_t> at(props.size());
        for (size_t p = 0; p < props.size(); ++p) { at[p] = stride; stride += (size_t)props[p].size; }
        const size_t chunk = 65536;
        std::vector<unsigned char> buf(chunk * stride);
        for (int64_t i0 = 0; i0 < n && rc == F4L_OK; i0 += (int64_t)chunk) {
            const size_t m = (size_t)std::min<int64_t>((int64_t)chunk, n - i0);
            if (fread(buf.data(), stride, m, fp) != m) { rc = F4L_EINVAL; break; }
            for (size_t j = 0; j < m; ++j) {
                const unsigned char *r = buf.data() + j * stride;
                const size_t i = (size_t)i0 + j;
                c.xyz[3 * i] = (float)load_value(r + at[(size_t)ix], props[(size_t)ix].type, swap);
                c.xyz[3 * i + 1] = (float)load_value(r + at[(size_t)iy], props[(size_t)iy].type, swap);
                c.xyz[3 * i + 2] = (float)load_value(r + at[(size_t)iz], props[(size_t)iz].type, swap);
                if (c.has_rgb) {
                    c.rgb[3 * i] = (uint8_t)load_value(r + at[(size_t)ir], props[(size_t)ir].type, swap);
                    c.rgb[3 * i + 1] = (uint8_t)load_value(r + at[(size_t)ig], props[(size_t)ig].type, swap);
                    c.rgb[3 * i + 2] = (uint8_t)load_value(r + at[(size_t)ib], props[(size_t)ib].type, swap);
                }
            }
        }
    } else rc = F4L_EINVAL;
    fclose(fp);
    return rc;
}
// Binary little-endian PLY with float x y z and, when present, uchar red green blue (PLYWriter::write(..., binary = true,
// use_camera = false) of a PointXYZRGB cloud, pcd_tiling.cpp:263-268).  `vertex`: packed records of 12 or 15 bytes.
static int write_ply(const std::string &path, const unsigned char *vertex, int64_t n, bool has_rgb) {
    FILE *fp = fopen(path.c_str(), "wb");
    if (!fp) return F4L_EINVAL;
    std::string head = "ply\nformat binary_little_endian 1.0\nelement vertex " + std::to_string(n) + "\nproperty float x\nproperty float y\nproperty float z\n";
    if (has_rgb) head += "property uchar red\nproperty uchar green\nproperty uchar blue\n";
    head += "end_header\n";
    bool ok = fwrite(head.data(), 1, head.size(), fp) == head.size();
    const size_t bytes = (size_t)n * (has_rgb ? 15 : 12);
    ok = ok && (bytes == 0 || fwrite(vertex, 1, bytes, fp) == bytes);
    ok = (fclose(fp) == 0) && ok;
    return ok ? F4L_OK : F4L_EINVAL;
}

// ---- device side ---------------------------------------------------------------------------------------------------------------
struct Box { float lo[3], hi[3]; };
struct InBox {  // pcl::CropBox with min / max (pcd_tiling.cpp:104-116): keeps lo <= p <= hi on every axis, in float32
    const float *xyz;
    Box b;
    __device__ __forceinline__ bool operator()(const int32_t &i) const {
        const float x = xyz[3 * (int64_t)i], y = xyz[3 * (int64_t)i + 1], z = xyz[3 * (int64_t)i + 2];
        return x >= b.lo[0] && x <= b.hi[0] && y >= b.lo[1] && y <= b.hi[1] && z >= b.lo[2] && z <= b.hi[2];
    }
};
#define TL_FOR(i, n) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)(n); i += (int64_t)gridDim.x * blockDim.x)
__device__ __forceinline__ unsigned int f2ord(float f) {
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float ord2f(unsigned int o) {
    const unsigned int u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
__global__ void bbox_kernel(const float *__restrict__ xyz, int64_t n, unsigned int *bb) {  // bb[0..2] min, bb[3..5] max (ordered bits)
    unsigned int mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
    TL_FOR(i, n) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const unsigned int o = f2ord(xyz[3 * i + d]);
            mn[d] = o < mn[d] ? o : mn[d];
            mx[d] = o > mx[d] ? o : mx[d];
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned int a = (unsigned int)__shfl_xor((int)mn[d], m, 64), b = (unsigned int)__shfl_xor((int)mx[d], m, 64);
            mn[d] = a < mn[d] ? a : mn[d];
            mx[d] = b > mx[d] ? b : mx[d];
        }
    if (lane_id() == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { atomicMin(&bb[d], mn[d]); atomicMax(&bb[3 + d], mx[d]); }
    }
}
// how many points of the cloud lie in each of two boxes (the two halves of a node: what decides the recursion, :247-253)
__global__ void count2_kernel(const float *__restrict__ xyz, int64_t n, Box a, Box b, unsigned long long *cnt) {
    int ca = 0, cb = 0;
    TL_FOR(i, n) {
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        ca += (x >= a.lo[0] && x <= a.hi[0] && y >= a.lo[1] && y <= a.hi[1] && z >= a.lo[2] && z <= a.hi[2]) ? 1 : 0;
        cb += (x >= b.lo[0] && x <= b.hi[0] && y >= b.lo[1] && y <= b.hi[1] && z >= b.lo[2] && z <= b.hi[2]) ? 1 : 0;
    }
    ca = wave_sum(ca); cb = wave_sum(cb);
    if (lane_id() == 0) {
        if (ca) atomicAdd(&cnt[0], (unsigned long long)ca);
        if (cb) atomicAdd(&cnt[1], (unsigned long long)cb);
    }
}
// the selected vertices as the PLY writer's records: float x y z [+ uchar r g b], packed
__global__ void gather_vertex_kernel(const float *__restrict__ xyz, const uint8_t *__restrict__ rgb, const int32_t *__restrict__ idx, int64_t m,
                                     unsigned char *__restrict__ out) {
    const int rec = rgb ? 15 : 12;
    TL_FOR(t, m) {
        const int64_t i = idx[t];
        unsigned char *o = out + t * rec;
        const float v[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        memcpy(o, v, 12);
        if (rgb) { o[12] = rgb[3 * i]; o[13] = rgb[3 * i + 1]; o[14] = rgb[3 * i + 2]; }
    }
}
__global__ void gather_cloud_kernel(const float *__restrict__ xyz, const uint8_t *__restrict__ rgb, const int32_t *__restrict__ idx, int64_t m,
                                    float *__restrict__ xyz_out, uint8_t *__restrict__ rgb_out) {
    TL_FOR(t, m) {
        const int64_t i = idx[t];
#pragma unroll
        for (int d = 0; d < 3; ++d) xyz_out[3 * t + d] = xyz[3 * i + d];
        if (rgb) {
#pragma unroll
            for (int d = 0; d < 3; ++d) rgb_out[3 * t + d] = rgb[3 * i + d];
        }
    }
}
// the voxel filter's outputs: centroids (double, f4l_voxel_downsample) as float32, colours averaged like the coordinates -- the
// exact integer sums over the voxel's points divided in double and truncated when packed back into the rgb field
__global__ void centroid_f32_kernel(const double *__restrict__ c, int64_t m, float *__restrict__ out) {
    TL_FOR(t, 3 * m) out[t] = (float)c[t];
}
__global__ void colour_sum_kernel(const uint8_t *__restrict__ rgb, const int32_t *__restrict__ voxel_of_point, int64_t n, unsigned int *__restrict__ sums) {
    TL_FOR(i, n) {
        const int64_t v = voxel_of_point[i];
#pragma unroll
        for (int d = 0; d < 3; ++d) atomicAdd(&sums[3 * v + d], (unsigned int)rgb[3 * i + d]);
    }
}
__global__ void colour_mean_kernel(const unsigned int *__restrict__ sums, const int32_t *__restrict__ count, int64_t m, uint8_t *__restrict__ rgb_out) {
    TL_FOR(t, 3 * m) rgb_out[t] = (uint8_t)((double)sums[t] / (double)count[t / 3]);
}
__global__ void second_d2_kernel(const double *__restrict__ d2, int64_t n, double *__restrict__ out) {
    TL_FOR(i, n) out[i] = (double)(float)d2[2 * i + 1];  // (pointNKNSquaredDistance is float, :41-49)
}

struct DevCloud {
    float *xyz = nullptr;
    uint8_t *rgb = nullptr;
    int64_t n = 0;
    void release() {
        if (xyz) (void)hipFree(xyz);
        if (rgb) (void)hipFree(rgb);
        xyz = nullptr; rgb = nullptr; n = 0;
    }
};
struct Ctx {
    hipStream_t st;
    int32_t *idx = nullptr;           // [cap] selected indices
    unsigned char *vertex = nullptr;  // [cap][15] staging for a leaf's records
    void *tmp = nullptr;              // rocPRIM scratch
    size_t tmp_bytes = 0;
    unsigned long long *cnt = nullptr;  // device counters [4] + selected count
    unsigned long long *cnt_host = nullptr;  // pinned mirror
    std::vector<unsigned char> host;  // a leaf's records on the host
    int64_t cap = 0;
    void release() {
        if (idx) (void)hipFree(idx);
        if (vertex) (void)hipFree(vertex);
        if (tmp) (void)hipFree(tmp);
        if (cnt) (void)hipFree(cnt);
        if (cnt_host) (void)hipHostFree(cnt_host);
        idx = nullptr; vertex = nullptr; tmp = nullptr; cnt = nullptr; cnt_host = nullptr;
    }
};
static inline dim3 grid_for(int64_t n) {
    const int64_t b = (n + 255) / 256;
    return dim3((unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b)));
}
static int upload(const HostCloud &h, DevCloud &d, hipStream_t st) {
    d.n = h.n;
    if (h.n == 0) return F4L_OK;
    F4L_HIP_CHECK(hipMalloc((void **)&d.xyz, (size_t)h.n * 12));
    F4L_HIP_CHECK(hipMemcpyAsync(d.xyz, h.xyz.data(), (size_t)h.n * 12, hipMemcpyHostToDevice, st));
    if (h.has_rgb) {
        F4L_HIP_CHECK(hipMalloc((void **)&d.rgb, (size_t)h.n * 3));
        F4L_HIP_CHECK(hipMemcpyAsync(d.rgb, h.rgb.data(), (size_t)h.n * 3, hipMemcpyHostToDevice, st));
    }
    F4L_HIP_CHECK(hipStreamSynchronize(st));
    return F4L_OK;
}
static int bbox(const DevCloud &c, Ctx &x, float (&lo)[3], float (&hi)[3]) {
    unsigned int init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u}, out[6];
    unsigned int *bb = reinterpret_cast<unsigned int *>(x.cnt);
    F4L_HIP_CHECK(hipMemcpyAsync(bb, init, sizeof init, hipMemcpyHostToDevice, x.st));
    hipLaunchKernelGGL(bbox_kernel, grid_for(c.n), dim3(256), 0, x.st, c.xyz, c.n, bb);
    F4L_LAUNCH_CHECK();
    F4L_HIP_CHECK(hipMemcpyAsync(out, bb, sizeof out, hipMemcpyDeviceToHost, x.st));
    F4L_HIP_CHECK(hipStreamSynchronize(x.st));
    for (int d = 0; d < 3; ++d) { lo[d] = ord2f(out[d]); hi[d] = ord2f(out[3 + d]); }
    return F4L_OK;
}
// indices of the cloud's points inside the box, ascending (x.idx), and their number
static int select_box(const DevCloud &c, const Box &b, Ctx &x, int64_t &m) {
    m = 0;
    if (c.n == 0) return F4L_OK;
    size_t tb = x.tmp_bytes;
    InBox pred{c.xyz, b};
    F4L_HIP_CHECK(rocprim::select(x.tmp, tb, rocprim::counting_iterator<int32_t>(0), x.idx, x.cnt + 4, (size_t)c.n, pred, x.st, false));
    F4L_HIP_CHECK(hipMemcpyAsync(x.cnt_host + 4, x.cnt + 4, 8, hipMemcpyDeviceToHost, x.st));
    F4L_HIP_CHECK(hipStreamSynchronize(x.st));
    m = (int64_t)x.cnt_host[4];
    return F4L_OK;
}
static int crop_cloud(DevCloud &c, const Box &b, Ctx &x) {  // filter_based_on_bb (:104-116), in place
    int64_t m = 0;
    int rc = select_box(c, b, x, m);
    if (rc != F4L_OK) return rc;
    if (m == c.n) return F4L_OK;
    DevCloud out;
    out.n = m;
    if (m > 0) {
        F4L_HIP_CHECK(hipMalloc((void **)&out.xyz, (size_t)m * 12));
        if (c.rgb) F4L_HIP_CHECK(hipMalloc((void **)&out.rgb, (size_t)m * 3));
        hipLaunchKernelGGL(gather_cloud_kernel, grid_for(m), dim3(256), 0, x.st, c.xyz, c.rgb, x.idx, m, out.xyz, out.rgb);
        F4L_LAUNCH_CHECK();
        F4L_HIP_CHECK(hipStreamSynchronize(x.st));
    }
    const bool had_rgb = c.rgb != nullptr;
    c.release();
    c = out;
    if (m == 0 && had_rgb) c.rgb = nullptr;
    return F4L_OK;
}
// median_point_cloud_resolution (:37-54): sqrt of the upper median (index n / 2) of the float squared distance to the nearest other point
static int median_resolution(const DevCloud &c, Ctx &x, float &res) {
    if (c.n < 2) return F4L_EINVAL;
    const size_t ws_b = f4l_knn_workspace_bytes(c.n, 2);
    void *ws = nullptr;
    int32_t *idx = nullptr;
    double *d2 = nullptr, *second = nullptr, *out = nullptr;
    void *sel = nullptr;
    int rc = F4L_OK;
    auto done = [&]() {
        if (ws) (void)hipFree(ws);
        if (idx) (void)hipFree(idx);
        if (d2) (void)hipFree(d2);
        if (second) (void)hipFree(second);
        if (out) (void)hipFree(out);
        if (sel) (void)hipFree(sel);
    };
#define TL_TRY(call) do { if ((call) != hipSuccess) { (void)hipGetLastError(); done(); return F4L_EHIP; } } while (0)
    TL_TRY(hipMalloc(&ws, ws_b ? ws_b : 1));
    TL_TRY(hipMalloc((void **)&idx, (size_t)c.n * 8));
    TL_TRY(hipMalloc((void **)&d2, (size_t)c.n * 16));
    TL_TRY(hipMalloc((void **)&second, (size_t)c.n * 8));
    TL_TRY(hipMalloc((void **)&out, 16));
    TL_TRY(hipMalloc(&sel, select_workspace_bytes()));
    rc = f4l_knn(c.xyz, c.n, 2, idx, d2, ws, ws_b, x.st);
    if (rc == F4L_OK) {
        hipLaunchKernelGGL(second_d2_kernel, grid_for(c.n), dim3(256), 0, x.st, (const double *)d2, c.n, second);
        const int64_t rank = c.n / 2;
        rc = select_ranks_f64(second, c.n, 1, 1, &rank, out, sel, x.st);
    }
    double v = 0.0;
    if (rc == F4L_OK) {
        TL_TRY(hipMemcpyAsync(&v, out, 8, hipMemcpyDeviceToHost, x.st));
        TL_TRY(hipStreamSynchronize(x.st));
        res = std::sqrt((float)v);
    }
#undef TL_TRY
    done();
    return rc;
}
// voxel_grid_filter (:118-227) in one piece (the reference splits into octants beyond 2^31 leaves because pcl::VoxelGrid indexes
// cells with int32; the keys of f4l_voxel_downsample are 64 bit)
static int voxel_grid(DevCloud &c, float leaf, Ctx &x) {
    if (c.n == 0) return F4L_OK;
    const size_t ws_b = f4l_voxel_downsample_workspace_bytes(c.n);
    void *ws = nullptr;
    double *pts = nullptr;
    int32_t *count = nullptr, *vop = nullptr;
    unsigned int *sums = nullptr;
    DevCloud out;
    int rc = F4L_OK;
    auto done = [&]() {
        if (ws) (void)hipFree(ws);
        if (pts) (void)hipFree(pts);
        if (count) (void)hipFree(count);
        if (vop) (void)hipFree(vop);
        if (sums) (void)hipFree(sums);
    };
#define TL_TRY(call) do { if ((call) != hipSuccess) { (void)hipGetLastError(); done(); out.release(); return F4L_EHIP; } } while (0)
    TL_TRY(hipMalloc(&ws, ws_b ? ws_b : 1));
    TL_TRY(hipMalloc((void **)&pts, (size_t)c.n * 24));
    TL_TRY(hipMalloc((void **)&count, (size_t)c.n * 4));
    TL_TRY(hipMalloc((void **)&vop, (size_t)c.n * 4));
    int64_t m = 0;
    rc = f4l_voxel_downsample(c.xyz, c.n, (double)leaf, F4L_VOXEL_PCL, pts, count, vop, &m, ws, ws_b, x.st);
    if (rc != F4L_OK) { done(); return rc; }
    out.n = m;
    TL_TRY(hipMalloc((void **)&out.xyz, (size_t)(m > 0 ? m : 1) * 12));
    hipLaunchKernelGGL(centroid_f32_kernel, grid_for(3 * m), dim3(256), 0, x.st, (const double *)pts, m, out.xyz);
    if (c.rgb) {
        TL_TRY(hipMalloc((void **)&out.rgb, (size_t)(m > 0 ? m : 1) * 3));
        TL_TRY(hipMalloc((void **)&sums, (size_t)(m > 0 ? m : 1) * 12));
        TL_TRY(hipMemsetAsync(sums, 0, (size_t)(m > 0 ? m : 1) * 12, x.st));
        hipLaunchKernelGGL(colour_sum_kernel, grid_for(c.n), dim3(256), 0, x.st, (const uint8_t *)c.rgb, (const int32_t *)vop, c.n, sums);
        hipLaunchKernelGGL(colour_mean_kernel, grid_for(3 * m), dim3(256), 0, x.st, (const unsigned int *)sums, (const int32_t *)count, m, out.rgb);
    }
    TL_TRY(hipGetLastError());
    TL_TRY(hipStreamSynchronize(x.st));
#undef TL_TRY
    done();
    c.release();
    c = out;
    return F4L_OK;
}

static inline float f32(double v) { return (float)v; }
// One halving step of split_point_clouds_into_tiles (:276-655): two (tile box, padded box) pairs, upper half first (the order the
// reference recurses in).  float32 boxes; the EPS terms enter in double and are rounded away again when stored, exactly as
// `float = float - float / 2 - 1e-9` does.
static void split_boxes(const Box &b, int direction, Box (&tile)[2], Box (&pad)[2]) {
    static const int UV[3][2] = {{1, 2}, {0, 2}, {0, 1}};
    const int u = UV[direction][0], v = UV[direction][1];
    const float side_u = b.hi[u] - b.lo[u], side_v = b.hi[v] - b.lo[v];
    const int s = side_u > side_v ? u : v, o = side_u > side_v ? v : u;
    const float half = (b.hi[s] - b.lo[s]) / 2.0f;
    const float cut = b.hi[s] - half;
    for (int k = 0; k < 2; ++k) {
        tile[k] = b;
        pad[k] = b;
        if (k == 0) {  // upper
            tile[k].lo[s] = f32((double)cut - EPS);
            pad[k].lo[s] = f32((double)cut - EPS - PAD);
            pad[k].hi[s] = f32((double)b.hi[s] + PAD);
        } else {
            tile[k].hi[s] = f32((double)cut + EPS);
            pad[k].lo[s] = f32((double)b.lo[s] - PAD);
            // (one branch of the reference subtracts EPS here instead of adding it: projection along z, split along x, :570)
            const double sign = (direction == 2 && s == 0) ? -1.0 : 1.0;
            pad[k].hi[s] = f32((double)cut + sign * EPS + PAD);
        }
        pad[k].lo[o] = f32((double)b.lo[o] - PAD);
        pad[k].hi[o] = f32((double)b.hi[o] + PAD);
    }
}
static int write_leaf(const DevCloud &c, const Box &b, const std::string &path, Ctx &x) {
    int64_t m = 0;
    int rc = select_box(c, b, x, m);
    if (rc != F4L_OK) return rc;
    const int rec = c.rgb ? 15 : 12;
    if (m > 0) {
        hipLaunchKernelGGL(gather_vertex_kernel, grid_for(m), dim3(256), 0, x.st, (const float *)c.xyz, (const uint8_t *)c.rgb, (const int32_t *)x.idx, m, x.vertex);
        F4L_LAUNCH_CHECK();
        try { x.host.resize((size_t)m * (size_t)rec); } catch (const std::bad_alloc &) { return F4L_ENOMEM; }
        F4L_HIP_CHECK(hipMemcpyAsync(x.host.data(), x.vertex, (size_t)m * (size_t)rec, hipMemcpyDeviceToHost, x.st));
        F4L_HIP_CHECK(hipStreamSynchronize(x.st));
    }
    return write_ply(path, x.host.data(), m, c.rgb != nullptr);
}
struct Job {
    const DevCloud *c1, *c2;
    int64_t max_pts;
    int direction;
    std::string save_dir;
    int counter = 0;
};
// split_point_clouds_into_tiles (:231-655): n1 / n2 = points of the two clouds in `b`
static int split(Job &j, const Box &b, const Box &padded, int64_t n1, int64_t n2, Ctx &x, int depth) {
    if (std::max(n1, n2) / j.max_pts + 1 == 1) {  // (:247-248) small enough: write it
        if (std::min(n1, n2) > 1) {               // (:253; the 1000-point floor is commented out in the reference)
            const std::string i = std::to_string(j.counter);
            int rc = write_leaf(*j.c1, b, j.save_dir + "/non_overlap/source_tile_" + i + ".ply", x);
            if (rc == F4L_OK) rc = write_leaf(*j.c2, b, j.save_dir + "/non_overlap/target_tile_" + i + ".ply", x);
            if (rc == F4L_OK) rc = write_leaf(*j.c1, padded, j.save_dir + "/overlap/source_tile_" + i + "_overlap.ply", x);
            if (rc == F4L_OK) rc = write_leaf(*j.c2, padded, j.save_dir + "/overlap/target_tile_" + i + "_overlap.ply", x);
            if (rc != F4L_OK) return rc;
            ++j.counter;
        }
        return F4L_OK;
    }
    // more than maxPointsPerTile coincident points: the box cannot be halved any further (the reference recurses until the stack overflows)
    if ((!(b.hi[0] - b.lo[0] > 0.f) && !(b.hi[1] - b.lo[1] > 0.f) && !(b.hi[2] - b.lo[2] > 0.f)) || depth > 4096) return F4L_EUNSUPPORTED;
    Box tile[2], pad[2];
    split_boxes(b, j.direction, tile, pad);
    F4L_HIP_CHECK(hipMemsetAsync(x.cnt, 0, 32, x.st));
    hipLaunchKernelGGL(count2_kernel, grid_for(j.c1->n), dim3(256), 0, x.st, (const float *)j.c1->xyz, j.c1->n, tile[0], tile[1], x.cnt);
    hipLaunchKernelGGL(count2_kernel, grid_for(j.c2->n), dim3(256), 0, x.st, (const float *)j.c2->xyz, j.c2->n, tile[0], tile[1], x.cnt + 2);
    F4L_LAUNCH_CHECK();
    F4L_HIP_CHECK(hipMemcpyAsync(x.cnt_host, x.cnt, 32, hipMemcpyDeviceToHost, x.st));
    F4L_HIP_CHECK(hipStreamSynchronize(x.st));
    const int64_t c1n[2] = {(int64_t)x.cnt_host[0], (int64_t)x.cnt_host[1]}, c2n[2] = {(int64_t)x.cnt_host[2], (int64_t)x.cnt_host[3]};
    for (int k = 0; k < 2; ++k) {
        const int rc = split(j, tile[k], pad[k], c1n[k], c2n[k], x, depth + 1);
        if (rc != F4L_OK) return rc;
    }
    return F4L_OK;
}
static bool is_file(const char *p) {
    struct stat s;
    return p && stat(p, &s) == 0 && S_ISREG(s.st_mode);
}
static void make_dir(const std::string &p) { (void)mkdir(p.c_str(), 0777); }  // (create_directory: no error when it exists, :800-809)
}  // namespace tiling
}  // namespace f4l

extern "C" int f4l_tile_point_clouds(const char *first_ply, const char *second_ply, int max_points_per_tile, int min_points_per_tile,
                                     int voxel_grid_flag, float voxel_grid_filter_size, float overlap_tiles, int projection_direction,
                                     const char *save_dir, int verbose, int32_t *n_tiles_host, void *stream) {
    using namespace f4l;
    using namespace f4l::tiling;
    (void)min_points_per_tile; (void)overlap_tiles;  // (accepted and unused, as in the reference: the pad is a hard-coded 20 m)
    if (!first_ply || !second_ply || !save_dir || max_points_per_tile < 1 || projection_direction < -1 || projection_direction > 2) return F4L_EINVAL;
    if (n_tiles_host) *n_tiles_host = -1;
    if (!is_file(first_ply) || !is_file(second_ply)) {  // (:735-738, 750-753: the reference prints and returns false)
        if (n_tiles_host) *n_tiles_host = -2;
        return F4L_OK;
    }
    hipStream_t st = (hipStream_t)stream;
    HostCloud h1, h2;
    int rc = read_ply(first_ply, h1);
    if (rc == F4L_OK) rc = read_ply(second_ply, h2);
    if (rc != F4L_OK) return rc;
    if (verbose) printf("Point cloud 1 read in! Number of Points: %lld\nPoint cloud 2 read in! Number of Points: %lld\n", (long long)h1.n, (long long)h2.n);
    if (h1.n == 0 || h2.n == 0) return F4L_EINVAL;
    if (h1.n > 0x7fffffffLL || h2.n > 0x7fffffffLL) return F4L_EUNSUPPORTED;
    DevCloud c1, c2;
    Ctx x;
    x.st = st;
    auto cleanup = [&](int code) { c1.release(); c2.release(); x.release(); return code; };
    rc = upload(h1, c1, st);
    if (rc == F4L_OK) rc = upload(h2, c2, st);
    if (rc != F4L_OK) return cleanup(rc);
    { HostCloud e1, e2; std::swap(h1, e1); std::swap(h2, e2); }  // (the host copies are not needed any more)
    x.cap = std::max(c1.n, c2.n);
    {
        size_t tb = 0;
        InBox pred{nullptr, Box{}};
        if (rocprim::select(nullptr, tb, rocprim::counting_iterator<int32_t>(0), (int32_t *)nullptr, (unsigned long long *)nullptr, (size_t)x.cap, pred, st, false) != hipSuccess)
            return cleanup(F4L_EHIP);
        x.tmp_bytes = tb ? tb : 1;
        if (hipMalloc(&x.tmp, x.tmp_bytes) != hipSuccess || hipMalloc((void **)&x.idx, (size_t)x.cap * 4) != hipSuccess ||
            hipMalloc((void **)&x.vertex, (size_t)x.cap * 15) != hipSuccess || hipMalloc((void **)&x.cnt, 64) != hipSuccess ||
            hipHostMalloc((void **)&x.cnt_host, 64, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return cleanup(F4L_EHIP);
        }
    }
    // overlap of the two bounding boxes and the area of its three faces (:73-102, 763-772)
    float lo1[3], hi1[3], lo2[3], hi2[3];
    rc = bbox(c1, x, lo1, hi1);
    if (rc == F4L_OK) rc = bbox(c2, x, lo2, hi2);
    if (rc != F4L_OK) return cleanup(rc);
    Box root;
    for (int d = 0; d < 3; ++d) { root.lo[d] = std::max(lo1[d], lo2[d]); root.hi[d] = std::min(hi1[d], hi2[d]); }
    const float ext[3] = {root.hi[0] - root.lo[0], root.hi[1] - root.lo[1], root.hi[2] - root.lo[2]};
    const float area[3] = {ext[1] * ext[2], ext[0] * ext[2], ext[0] * ext[1]};
    rc = crop_cloud(c1, root, x);
    if (rc == F4L_OK) rc = crop_cloud(c2, root, x);
    if (rc != F4L_OK) return cleanup(rc);
    const std::string dir(save_dir);
    make_dir(dir); make_dir(dir + "/non_overlap"); make_dir(dir + "/overlap");
    if (voxel_grid_flag) {
        float leaf = voxel_grid_filter_size;
        if (leaf == 0.0f) {  // (:814-821) spacing of the smaller cloud
            rc = median_resolution(c1.n < c2.n ? c1 : c2, x, leaf);
            if (rc != F4L_OK) return cleanup(rc);
            if (verbose) printf("Size of the filter: %g m determined based on the median resolution!\n", (double)leaf);
        }
        rc = voxel_grid(c1, leaf, x);
        if (rc == F4L_OK) rc = voxel_grid(c2, leaf, x);
        if (rc != F4L_OK) return cleanup(rc);
        if (verbose) printf("%lld / %lld points remaining after voxel grid filter.\n", (long long)c1.n, (long long)c2.n);
    }
    int direction = projection_direction;
    if (direction == -1) {  // (:844-845) project along the axis whose face of the overlap box is largest (the first of equal ones)
        direction = 0;
        for (int d = 1; d < 3; ++d)
            if (area[d] > area[direction]) direction = d;
    }
    Job job;
    job.c1 = &c1; job.c2 = &c2; job.max_pts = max_points_per_tile; job.direction = direction; job.save_dir = dir;
    rc = split(job, root, root, c1.n, c2.n, x, 0);
    if (rc != F4L_OK) return cleanup(rc);
    if (verbose) printf("Spliting complete. %d patches saved per epoch.\n", job.counter);
    if (n_tiles_host) *n_tiles_host = job.counter;
    return cleanup(F4L_OK);
}

// resave_point_cloud (:662-707): both files re-written as binary PLY.  The reference loads the second cloud only when `verbose` is set
// (:692-697) and writes an empty cloud over it otherwise, and falls off the end without a return value; here both files are always
// read and rewritten.  *ok_host: 1 done, 0 a file is missing (the reference's `return false`).
extern "C" int f4l_resave_point_cloud(const char *first_ply, const char *second_ply, int verbose, int32_t *ok_host) {
    using namespace f4l;
    using namespace f4l::tiling;
    (void)verbose;
    if (!first_ply || !second_ply) return F4L_EINVAL;
    if (ok_host) *ok_host = 0;
    if (!is_file(first_ply) || !is_file(second_ply)) return F4L_OK;
    for (const char *p : {first_ply, second_ply}) {
        HostCloud h;
        int rc = read_ply(p, h);
        if (rc != F4L_OK) return rc;
        const int rec = h.has_rgb ? 15 : 12;
        std::vector<unsigned char> v((size_t)h.n * (size_t)rec);
        for (int64_t i = 0; i < h.n; ++i) {
            memcpy(v.data() + (size_t)i * rec, &h.xyz[3 * (size_t)i], 12);
            if (h.has_rgb) memcpy(v.data() + (size_t)i * rec + 12, &h.rgb[3 * (size_t)i], 3);
        }
        rc = write_ply(p, v.data(), h.n, h.has_rgb);
        if (rc != F4L_OK) return rc;
    }
    if (ok_host) *ok_host = 1;
    return F4L_OK;
}
