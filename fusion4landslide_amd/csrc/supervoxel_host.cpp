// supervoxel_host.cpp -- the HOST code of the supervoxel partition (plain C++, no HIP): the label-identical sequential
// segmentation and the partition text writer.  Kept apart from the kernels so that it also builds under
// AddressSanitizer / UndefinedBehaviorSanitizer on the CPU (`make -C oracle asan`).
//
// The boundary-preserving segmentation (codelibrary/geometry/point_cloud/supervoxel_segmentation.h:65-265) has a fusion
// pass that is sequential and ORDER DEPENDENT (the visiting order of the representatives, a mutable adjacency, an early
// `break` when the count hits K), so a label-identical result requires replaying that order.  It is written here as a flat,
// allocation-free routine (pooled adjacency lists, SoA points) rather than the reference's Array<Array<int>> of heap
// vectors.  (The all-device variant that gives up label identity is supervoxel_gpu.hip.)
#include <algorithm>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <random>
#include <thread>
#include <vector>

#include "../../include/f4l.h"
#include "supervoxel_host.h"

namespace f4l {

namespace {

struct Segmenter {
    const float *xyz;
    const double *nrm;
    const int32_t *knn;
    int32_t n;
    int k;
    double resolution;

    // VCCS metric, supervoxel.cpp:33-37
    inline double metric(int32_t a, int32_t b) const {
        const float *pa = xyz + 3 * (size_t)a, *pb = xyz + 3 * (size_t)b;
        const double *na = nrm + 3 * (size_t)a, *nb = nrm + 3 * (size_t)b;
        const double dot = na[0] * nb[0] + na[1] * nb[1] + na[2] * nb[2];
        const double t1 = (double)pa[0] - pb[0], t2 = (double)pa[1] - pb[1], t3 = (double)pa[2] - pb[2];
        return 1.0 - std::fabs(dot) + std::sqrt(t1 * t1 + t2 * t2 + t3 * t3) / resolution * 0.4;
    }
};

inline int32_t find_root(int32_t *parent, int32_t i) {  // path halving, disjoint_set.h:59-66
    while (i != parent[i]) {
        parent[i] = parent[parent[i]];
        i = parent[i];
    }
    return i;
}

// K = number of occupied cells of the resolution grid anchored at the bbox minimum (grid_sample.h:48-68)
int32_t occupied_cells(const float *xyz, int32_t n, double resolution) {
    double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (int32_t i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) {
            const double v = xyz[3 * (size_t)i + d];
            mn[d] = std::min(mn[d], v);
            mx[d] = std::max(mx[d], v);
        }
    int size[3];
    for (int d = 0; d < 3; ++d) size[d] = (int)((mx[d] - mn[d]) / resolution + 1);
    std::vector<uint64_t> keys((size_t)n);
    for (int32_t i = 0; i < n; ++i) {
        int c[3];
        for (int d = 0; d < 3; ++d) {
            c[d] = (int)(((double)xyz[3 * (size_t)i + d] - mn[d]) / resolution);
            c[d] = std::min(std::max(c[d], 0), size[d] - 1);
        }
        keys[(size_t)i] = ((uint64_t)c[0] * (uint64_t)size[1] + (uint64_t)c[1]) * (uint64_t)size[2] + (uint64_t)c[2];
    }
    std::sort(keys.begin(), keys.end());
    return (int32_t)(std::unique(keys.begin(), keys.end()) - keys.begin());
}

}  // namespace

int segment_host(const float *xyz, const double *nrm, const int32_t *knn, int64_t n64, int k, double resolution,
                 int32_t *labels, const SegmentAssist &assist) {
    if (n64 <= 0 || n64 > 0x7fffffffLL || k < 1 || !(resolution > 0.0)) return F4L_EINVAL;
    const int32_t n = (int32_t)n64;
    Segmenter sg{xyz, nrm, knn, n, k, resolution};
    const int32_t n_target = occupied_cells(xyz, n, resolution);

    std::vector<int32_t> parent((size_t)n), reps((size_t)n), sizes((size_t)n, 1), queue((size_t)n);
    std::vector<uint8_t> visited((size_t)n, 0);
    std::vector<double> dis((size_t)n);
    // adjacency: (offset, length) into a growing pool; the initial lists alias the kNN table itself
    std::vector<int64_t> adj_off((size_t)n);
    std::vector<int32_t> adj_len((size_t)n, k);
    std::vector<int32_t> pool;  // lists rewritten by the fusion pass; index = adj_off - n*k
    pool.reserve((size_t)n * 4);
    const int64_t knn_span = (int64_t)n * k;
    auto adj_ptr = [&](int32_t i) -> const int32_t * {
        return adj_off[(size_t)i] < knn_span ? knn + adj_off[(size_t)i] : pool.data() + (adj_off[(size_t)i] - knn_span);
    };
    for (int32_t i = 0; i < n; ++i) {
        parent[(size_t)i] = i;
        reps[(size_t)i] = i;
        adj_off[(size_t)i] = (int64_t)i * k;
    }

    const bool timing = getenv("F4L_SV_TIMING") != nullptr;
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto tsec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const auto t_start = tnow();
    // starting lambda: median over points of the smallest metric to a neighbour (:105-113, median.h:27-30)
    if (assist.dis0) std::memcpy(dis.data(), assist.dis0, (size_t)n * sizeof(double));
    else
        for (int32_t i = 0; i < n; ++i) {
            double best = DBL_MAX;
            for (int j = 0; j < k; ++j) {
                const int32_t q = knn[(size_t)i * k + j];
                if (q != i) best = std::min(best, sg.metric(i, q));
            }
            dis[(size_t)i] = best;
        }
    double lambda;
    {
        std::vector<double> tmp(dis);
        std::nth_element(tmp.begin(), tmp.begin() + n / 2, tmp.end());
        lambda = std::max(DBL_EPSILON, tmp[(size_t)(n / 2)]);
    }

    const auto t_lambda = tnow();
    // step 1 (:117-176): greedy fusion, lambda doubling until exactly n_target representatives remain
    int32_t n_reps = n, live = n;
    std::vector<int32_t> kept;
    // The reference's loop has no exit for a neighbour graph with more connected components than the target count (k = 4 on a strip:
    // the fuzz case that found it): lambda doubles for ever and no round absorbs anything any more -- the call never returns.  A loss
    // is at most sizes * metric <= n (1 + 0.4 diagonal / resolution); a round that absorbs nothing with lambda beyond that proves that
    // none ever will: reported (F4L_EUNSUPPORTED) instead of replayed.
    double loss_cap;
    {
        double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
        for (int32_t i = 0; i < n; ++i)
            for (int d = 0; d < 3; ++d) {
                const double v = xyz[3 * (size_t)i + d];
                mn[d] = std::min(mn[d], v);
                mx[d] = std::max(mx[d], v);
            }
        const double diag = std::sqrt((mx[0] - mn[0]) * (mx[0] - mn[0]) + (mx[1] - mn[1]) * (mx[1] - mn[1]) + (mx[2] - mn[2]) * (mx[2] - mn[2]));
        loss_cap = (double)n * (1.0 + 0.4 * diag / resolution) * 1.0001;
    }
    for (;; lambda *= 2.0) {
        if (n_reps <= 1) break;
        const int32_t live_before = live;
        if (pool.size() > (size_t)n * 16) {  // compact the pool: keep only lists still referenced
            std::vector<int32_t> fresh;
            fresh.reserve((size_t)n * 4);
            for (int32_t s = 0; s < n_reps; ++s) {
                const int32_t i = reps[(size_t)s];
                if (adj_off[(size_t)i] >= knn_span && adj_len[(size_t)i] > 0) {
                    const int32_t *src = adj_ptr(i);
                    const int64_t at = (int64_t)fresh.size();
                    fresh.insert(fresh.end(), src, src + adj_len[(size_t)i]);
                    adj_off[(size_t)i] = knn_span + at;
                }
            }
            pool.swap(fresh);
        }
        for (int32_t s = 0; s < n_reps; ++s) {
            const int32_t i = reps[(size_t)s];
            // the visiting order is known in advance and jumps through memory: request what the coming
            // representatives will touch first (their list header, then the list itself and their own record)
            if (s + 16 < n_reps) {
                const int32_t f = reps[(size_t)s + 16];
                __builtin_prefetch(&adj_off[(size_t)f]);
                __builtin_prefetch(&adj_len[(size_t)f]);
                __builtin_prefetch(&xyz[3 * (size_t)f]);
                __builtin_prefetch(&nrm[3 * (size_t)f]);
            }
            if (s + 8 < n_reps) {
                const int32_t f = reps[(size_t)s + 8];
                __builtin_prefetch(adj_ptr(f));
                __builtin_prefetch(adj_ptr(f) + 16);
                __builtin_prefetch(&sizes[(size_t)f]);
            }
            if (adj_len[(size_t)i] == 0) continue;
            visited[(size_t)i] = 1;
            int32_t front = 0, back = 1;
            queue[(size_t)front++] = i;
            {
                const int32_t *al = adj_ptr(i);
                for (int32_t a = 0; a < adj_len[(size_t)i]; ++a) {
                    const int32_t j = find_root(parent.data(), al[a]);
                    if (!visited[(size_t)j]) {
                        visited[(size_t)j] = 1;
                        queue[(size_t)back++] = j;
                    }
                }
            }
            kept.clear();
            while (front < back) {
                const int32_t j = queue[(size_t)front++];
                const double loss = sizes[(size_t)j] * sg.metric(i, j);
                const double improvement = lambda - loss;
                if (improvement > 0.0) {
                    parent[(size_t)j] = i;  // Link(j -> i), disjoint_set.h:77-85
                    sizes[(size_t)i] += sizes[(size_t)j];
                    const int32_t *al = adj_ptr(j);
                    for (int32_t a = 0; a < adj_len[(size_t)j]; ++a) {
                        const int32_t q = find_root(parent.data(), al[a]);
                        if (!visited[(size_t)q]) {
                            visited[(size_t)q] = 1;
                            queue[(size_t)back++] = q;
                        }
                    }
                    adj_len[(size_t)j] = 0;
                    if (--live == n_target) break;
                } else {
                    kept.push_back(j);
                }
            }
            adj_off[(size_t)i] = knn_span + (int64_t)pool.size();
            adj_len[(size_t)i] = (int32_t)kept.size();
            pool.insert(pool.end(), kept.begin(), kept.end());
            for (int32_t a = 0; a < back; ++a) visited[(size_t)queue[(size_t)a]] = 0;
            if (live == n_target) break;
        }
        int32_t m = 0;
        for (int32_t s = 0; s < n_reps; ++s) {
            const int32_t i = reps[(size_t)s];
            if (find_root(parent.data(), i) == i) reps[(size_t)m++] = i;
        }
        n_reps = m;
        if (live == live_before && !(lambda <= loss_cap)) return F4L_EUNSUPPORTED;  // (also a NaN lambda: never)
        live = m;
        if (n_reps == n_target) break;
    }

    const auto t_fuse = tnow();
    for (int32_t i = 0; i < n; ++i) labels[i] = find_root(parent.data(), i);  // :179-182

    // step 2 (:186-237): boundary exchange with a FIFO of points whose neighbourhood straddles two labels
    std::vector<uint8_t> flag;  // (assisted) points with a neighbour of another label: only they start pushes below
    if (assist.boundary) {
        flag.resize((size_t)n);
        assist.boundary(labels, flag.data(), dis.data(), assist.ctx);
    } else
        for (int32_t i = 0; i < n; ++i) dis[(size_t)i] = sg.metric(i, labels[i]);
    std::vector<int32_t> fifo((size_t)n);
    std::vector<uint8_t> in_q((size_t)n, 0);
    int64_t head = 0, tail = 0, count = 0;
    auto push = [&](int32_t v) {
        fifo[(size_t)tail] = v;
        tail = tail + 1 == n ? 0 : tail + 1;
        ++count;
        in_q[(size_t)v] = 1;
    };
    for (int32_t i = 0; i < n; ++i) {
        if (!flag.empty() && !flag[(size_t)i]) continue;  // no neighbour of another label: the scan below pushes nothing
        for (int j = 0; j < k; ++j) {
            const int32_t q = knn[(size_t)i * k + j];
            if (labels[i] != labels[q]) {
                if (!in_q[(size_t)i]) push(i);
                if (!in_q[(size_t)q]) push(q);
            }
        }
    }
    while (count > 0) {
        const int32_t i = fifo[(size_t)head];
        head = head + 1 == n ? 0 : head + 1;
        --count;
        in_q[(size_t)i] = 0;
        bool change = false;
        for (int j = 0; j < k; ++j) {
            const int32_t q = knn[(size_t)i * k + j];
            const int32_t a = labels[i], b = labels[q];
            if (a == b) continue;
            const double d = sg.metric(i, b);
            if (d < dis[(size_t)i]) {
                labels[i] = b;
                dis[(size_t)i] = d;
                change = true;
            }
        }
        if (change)
            for (int j = 0; j < k; ++j) {
                const int32_t q = knn[(size_t)i * k + j];
                if (labels[i] != labels[q] && !in_q[(size_t)q]) push(q);
            }
    }

    const auto t_refine = tnow();
    if (timing) fprintf(stderr, "[sv timing] n=%d lambda0 %.3f s, fusion %.3f s, refine %.3f s\n", n, tsec(t_start, t_lambda), tsec(t_lambda, t_fuse), tsec(t_fuse, t_refine));
    // step 3 (:241-247): relabel 0..K-1 in representative order
    std::vector<int32_t> &map = queue;  // reuse
    for (int32_t s = 0; s < n_reps; ++s) map[(size_t)reps[(size_t)s]] = s;
    for (int32_t i = 0; i < n; ++i) labels[i] = map[(size_t)labels[i]];
    return n_reps;
}

}  // namespace f4l

extern "C" int f4l_supervoxel_segment_host(const float *xyz_host, const double *normals_host, const int32_t *knn_host,
                                           int64_t n, int k, double resolution, int32_t *labels_host,
                                           int32_t *n_supervoxels_host) {
    if (!xyz_host || !normals_host || !knn_host || !labels_host) return F4L_EINVAL;
    int rc;
    try {
        rc = f4l::segment_host(xyz_host, normals_host, knn_host, n, k, resolution, labels_host);
    } catch (const std::bad_alloc &) {
        return F4L_ENOMEM;
    }
    if (rc < 0) return rc;
    if (n_supervoxels_host) *n_supervoxels_host = rc;
    return F4L_OK;
}

// Partition text file `x y z r g b label` exactly as the reference writes it (supervoxel.cpp:45-64 ->
// codelibrary/geometry/io/xyz_io.h:192-221): 12 significant digits, one random colour per supervoxel drawn from
// a default-seeded std::mt19937.  `load_partition` re-reads column 6 (src/coarse_to_fine_matching_base.py:1275).
namespace f4l {
// `out << std::setprecision(12) << (double)f` = printf("%.12g"): twelve significant digits of the exact value, to nearest (ties to
// even), trailing zeros dropped.  For 1 <= |f| < 10^12 -- coordinates -- the digits are rint(|f| * 10^(11 - e)), e the decimal
// exponent: a float32 times a power of ten up to 10^11 is exact in double (24 + 26 significant bits).  Everything else (zeros,
// fractions, huge values, non-finite ones) goes through snprintf.  1.84 s -> 0.2 s per million points (round 5).
static inline char *put_g12(char *p, float f) {
    const double v = (double)f, a = std::fabs(v);
    if (!(a >= 1.0 && a < 1e12)) return p + snprintf(p, 48, "%.12g", v);
    static const double P10[13] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12};
    int e = 0;
    while (a >= P10[e + 1]) ++e;  // 10^e <= a < 10^(e + 1), e <= 11
    double r = std::rint(a * P10[11 - e]);  // twelve digits (exact product)
    if (r >= 1e12) { r = 1e11; ++e; }       // rounded up into a thirteenth digit: 999999999999.6 -> 1e12
    if (e >= 12) return p + snprintf(p, 48, "%.12g", v);  // (printf switches to the exponent form there)
    unsigned long long u = (unsigned long long)r;
    char d[12];
    for (int i = 11; i >= 0; --i) { d[i] = (char)('0' + u % 10ULL); u /= 10ULL; }
    if (std::signbit(v)) *p++ = '-';
    int last = 11;
    while (last > e && d[last] == '0') --last;  // trailing zeros of the fraction go
    for (int i = 0; i <= e; ++i) *p++ = d[i];
    if (last > e) {
        *p++ = '.';
        for (int i = e + 1; i <= last; ++i) *p++ = d[i];
    }
    return p;
}
static inline char *put_uint(char *p, unsigned int v) {
    char t[12];
    int n = 0;
    do { t[n++] = (char)('0' + v % 10u); v /= 10u; } while (v);
    while (n) *p++ = t[--n];
    return p;
}
}  // namespace f4l


namespace f4l {
// Formatting split over cores (round 6): rows are formatted in blocks of `block` rows by up to `threads` workers at a time, each into
// a buffer of its own, and the blocks are written in order -- the same bytes as one thread writes, at a multiple of its rate (a
// 1 M-point tile's partition file: 0.18 s on one core).  F4L_WRITER_THREADS caps the workers (default: the cores, at most 8).
static int writer_threads() {
    int t = (int)std::thread::hardware_concurrency();
    if (t < 1) t = 1;
    if (t > 8) t = 8;
    if (const char *e = getenv("F4L_WRITER_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) t = v; }
    return t;
}
// fmt(r0, r1, out) formats rows [r0, r1) at `out` and returns the end (or nullptr on bad input); bytes_per_row bounds a row.
template <class Fmt>
static int write_rows_parallel(FILE *fp, int64_t n, size_t bytes_per_row, const Fmt &fmt) {
    const int64_t block = 16384;
    const int T = (int)std::min<int64_t>(writer_threads(), (n + block - 1) / block > 0 ? (n + block - 1) / block : 1);
    std::vector<std::vector<char>> buf((size_t)T);
    std::vector<size_t> len((size_t)T);
    try {
        for (auto &b : buf) b.resize((size_t)block * bytes_per_row);
    } catch (const std::bad_alloc &) {
        return F4L_ENOMEM;
    }
    int rc = F4L_OK;
    for (int64_t base = 0; base < n && rc == F4L_OK; base += block * T) {
        const int live = (int)std::min<int64_t>(T, (n - base + block - 1) / block);
        auto work = [&](int t) {
            const int64_t r0 = base + (int64_t)t * block, r1 = std::min<int64_t>(n, r0 + block);
            char *end = fmt(r0, r1, buf[(size_t)t].data());
            len[(size_t)t] = end ? (size_t)(end - buf[(size_t)t].data()) : (size_t)-1;
        };
        if (live == 1) work(0);
        else {
            std::vector<std::thread> th;
            th.reserve((size_t)live - 1);
            try {
                for (int t = 1; t < live; ++t) th.emplace_back(work, t);
            } catch (...) {  // (no more threads to be had: the rest on this one)
                for (int t = (int)th.size() + 1; t < live; ++t) work(t);
            }
            work(0);
            for (auto &x : th) x.join();
        }
        for (int t = 0; t < live && rc == F4L_OK; ++t) {
            if (len[(size_t)t] == (size_t)-1) rc = F4L_EINVAL;
            else if (fwrite(buf[(size_t)t].data(), 1, len[(size_t)t], fp) != len[(size_t)t]) rc = F4L_EINVAL;
        }
    }
    return rc;
}
}  // namespace f4l

extern "C" int f4l_write_partition_txt(const char *path, const float *xyz_host, const int32_t *labels_host, int64_t n,
                                       int32_t n_supervoxels) {
    if (!path || n < 0 || n_supervoxels < 0 || (n > 0 && (!xyz_host || !labels_host))) return F4L_EINVAL;
    std::vector<uint32_t> colour((size_t)n_supervoxels);
    std::mt19937 random;
    for (int32_t i = 0; i < n_supervoxels; ++i) colour[(size_t)i] = (uint32_t)random();
    FILE *fp = fopen(path, "wb");
    if (!fp) return F4L_EINVAL;
    int rc = f4l::write_rows_parallel(fp, n, 208, [&](int64_t i0, int64_t i1, char *p) -> char * {
        for (int64_t i = i0; i < i1; ++i) {
            const int32_t l = labels_host[i];
            if (l < 0 || l >= n_supervoxels) return nullptr;
            const uint32_t c = colour[(size_t)l];
            p = f4l::put_g12(p, xyz_host[3 * i]); *p++ = ' ';
            p = f4l::put_g12(p, xyz_host[3 * i + 1]); *p++ = ' ';
            p = f4l::put_g12(p, xyz_host[3 * i + 2]); *p++ = ' ';
            p = f4l::put_uint(p, (c >> 16) & 0xffu); *p++ = ' ';
            p = f4l::put_uint(p, (c >> 8) & 0xffu); *p++ = ' ';
            p = f4l::put_uint(p, c & 0xffu); *p++ = ' ';
            p = f4l::put_uint(p, (unsigned int)l); *p++ = '\n';
        }
        return p;
    });
    if (fclose(fp) != 0 && rc == F4L_OK) rc = F4L_EINVAL;
    return rc;
}

// The result files of a tile -- `np.savetxt(path, rows, delimiter=" ", fmt="%.6f")` of save_process_dvf
// (src/coarse_to_fine_matching_base.py:3477-3537: four to eight files of up to a million rows per tile) -- byte for byte, without
// numpy's per-value Python formatting (2.4 s per million rows of six; this: 0.1 s).  A float32 times 10^6 is EXACT in double (24 + 14
// significant bits: 10^6 = 15625 * 2^6), so the six-decimal rounding printf performs on the exact value -- to nearest, ties to
// even -- is rint() of that product; values of 10^9 and beyond, infinities and NaNs go through snprintf / the names Python prints.
namespace f4l {
static inline char *put_fixed6(char *p, float f) {
    const double v = (double)f;
    if (!(std::fabs(v) < 1e9)) {  // large, inf or nan
        if (std::isnan(v)) { memcpy(p, "nan", 3); return p + 3; }
        if (std::isinf(v)) { const char *t = v < 0 ? "-inf" : "inf"; const size_t l = strlen(t); memcpy(p, t, l); return p + l; }
        return p + snprintf(p, 64, "%.6f", v);
    }
    if (std::signbit(v)) *p++ = '-';
    const double x = std::rint(std::fabs(v) * 1e6);  // exact product, printf's rounding
    unsigned long long u = (unsigned long long)x;
    const unsigned long long ip = u / 1000000ULL;
    unsigned int fp = (unsigned int)(u % 1000000ULL);
    char tmp[24];
    int n = 0;
    unsigned long long q = ip;
    do { tmp[n++] = (char)('0' + q % 10ULL); q /= 10ULL; } while (q);
    while (n) *p++ = tmp[--n];
    *p++ = '.';
    for (int d = 5; d >= 0; --d) { p[d] = (char)('0' + fp % 10u); fp /= 10u; }
    return p + 6;
}
}  // namespace f4l

extern "C" int f4l_write_rows_txt(const char *path, const float *rows_host, int64_t n, int ncols) {
    if (!path || n < 0 || ncols < 1 || (n > 0 && !rows_host)) return F4L_EINVAL;
    FILE *fp = fopen(path, "wb");
    if (!fp) return F4L_EINVAL;
    int rc = f4l::write_rows_parallel(fp, n, (size_t)ncols * 66 + 1, [&](int64_t r0, int64_t r1, char *p) -> char * {
        for (int64_t r = r0; r < r1; ++r) {
            const float *row = rows_host + (size_t)r * (size_t)ncols;
            for (int c = 0; c < ncols; ++c) {
                if (c) *p++ = ' ';
                p = f4l::put_fixed6(p, row[c]);
            }
            *p++ = '\n';
        }
        return p;
    });
    if (fclose(fp) != 0 && rc == F4L_OK) rc = F4L_EINVAL;
    return rc;
}
