// patch_grid.h -- a uniform grid over ONE target patch, resident in LDS, and the exact nearest-neighbour
// query against it.  Shared by icp_kernel (icp.hip) and nn_refine_kernel (patch_ops.hip).
//
// The reference answers "nearest target point within radius r" with an Open3D KD-tree per patch
// (utils/o3d_tools.py:46-50 -> registration_icp -> KDTreeFlann::SearchHybrid(p, r, 1), and
// src/coarse_to_fine_matching_base.py:70-80 for refine_dvfs_with_threshold).  A pointer-chasing tree is the
// wrong shape for a 64-wide wavefront; a patch is small (hundreds to a few thousand points) and the radius is
// known, so the whole patch is counting-sorted ONCE into cubic cells of edge h >= r that live in LDS for all
// ICP iterations:
//
//   tl[0..nt)      target points in cell order, {x, y, z, tag = original index : slot}  (patch-relative coordinates);
//   tl[nt]         a dummy point at infinity that idle lanes read
//   E[0..ncell]    packed uint16 prefix table: points of cell c are tl[E[c] .. E[c+1])
//
// Cells are numbered x-fastest, so the three x-neighbours of a cell form ONE contiguous run ("row") of tl; a
// query therefore scans at most 9 rows (3 y x 3 z), each a linear run of float4 LDS reads.  Every lane owns
// one query and walks its own rows; rows whose distance to the query already exceeds the best candidate are
// skipped.  The result is the exact minimiser of (d2, original index) among the targets with d2 < r^2: the same
// point a brute-force scan in index order (and the oracle's KD-tree) returns, independent of the order in
// which the counting sort happened to place points inside a cell.
//
// Patches that are dense relative to the radius get cells finer than it (grid_build: r/2 .. r/8 while the bounding box
// still holds >= `dens` points per cell); a query whose bound exceeds one cell then walks the (2 W + 1)^2 runs of a
// wider stencil (grid_nn_wide).  Callers usually know a bound well below the radius (the previous correspondence,
// re-measured), which narrows every run to the cells the bound reaches and drops runs beyond it.
//
// Cell shape (round 4).  A cell is h x h in (y, z) -- the stencil of rows is decided there -- but only hx = h / XS along x,
// XS = 1, 2, 4 or 8 (the finest the cell table holds): finer cells along the run do not change which rows a query walks, only how tightly the run's x-range
// [px - bound, px + bound] is cut (whole cells): at C4 (2.6 points per h-cell, bounds of ~4 cm against h = 10 cm) a query
// measures 6-7 candidates instead of 11.  And a patch that is FLAT (terrain: z-extent below 3/4 of the larger of its x- and
// y-extents) gets ONE layer of cells, columns unbounded in z: with cubic cells most of the bounding box's z layers are empty
// and still count against the cell table, which kept the table from being spent on x.  Both are exact for any cloud (z simply
// stops pruning in a column grid); F4L_ICP_DEBUG bits 512 / 1024 switch them off (cubic cells, the round-3 grid).
//
// Exactness of the stencil (3x3x3 cells when h >= r, (2 w + 1)^3 with w = ceil(bound / h) otherwise): cell
// coordinates are floor((x - min) / h) evaluated in floating point for
// targets and queries alike (a monotone function of x); h = r * (1 + 2^-7) leaves 7e-3 cells of slack for its
// rounding (float32 cell coordinates below 16384 are good to 3e-3), so |q - t| < r implies |cell(q) - cell(t)| <= 1 on every axis.
#pragma once
#include "f4l_device.h"

namespace f4l {

// ---- point records as staged in LDS ------------------------------------------------------------------
template <typename F> struct GridPt;
template <> struct __attribute__((aligned(16))) GridPt<float> { float x, y, z; unsigned int tag; };                // 16 B
// float64 search: the ORIGINAL float32 coordinates; the patch-relative double is (double)x - origin, exact, formed on use
// (16 B as well: the float64 mode keeps the LDS footprint, hence the occupancy, of the float32 mode)
template <> struct __attribute__((aligned(16))) GridPt<double> { float x, y, z; unsigned int tag; };              // 16 B

template <typename F> struct PatchGrid {
    F minx, miny, minz, h, inv_h;
    F inv_hx;        // cells along x (the direction of a run) are h / XS wide: inv_hx = XS * inv_h
    F inv_hz;        // inv_h, or 0 for a column grid (nz == 1: every z lands in layer 0)
    int nx, ny, nz;  // nx * ny * nz <= cell capacity
    int wmax;        // cells per side of the stencil that covers the radius the grid was built for (>= 1), in units of h
    int xs;          // XS: x-cells per h
    F ox, oy, oz;    // patch origin (only the float64 records need it: they hold absolute coordinates)
};

// Distance arithmetic against a record.  float32: records are patch relative, subtract directly.  float64: records hold
// the original (absolute) float32 coordinate; the query is shifted back by the origin ONCE (grid_query) and the
// subtraction (query + origin) - (double)x is then what the reference's absolute-coordinate arithmetic evaluates,
// rounded at ulp(|coordinate|) like there.
__device__ __forceinline__ float grid_query(float p, float) { return p; }
__device__ __forceinline__ double grid_query(double p, double o) { return p + o; }
__device__ __forceinline__ float grid_coord(float v, float) { return v; }       // float32 record field -> as is
__device__ __forceinline__ double grid_coord(float v, double) { return (double)v; }  // float64 mode: widen

// patch-relative coordinates of a record
__device__ __forceinline__ void grid_rel(const PatchGrid<float> &, const GridPt<float> &q, float &x, float &y, float &z) {
    x = q.x; y = q.y; z = q.z;
}
__device__ __forceinline__ void grid_rel(const PatchGrid<double> &g, const GridPt<double> &q, double &x, double &y, double &z) {
    x = (double)q.x - g.ox; y = (double)q.y - g.oy; z = (double)q.z - g.oz;
}

__device__ __forceinline__ float grid_uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ double grid_uniform(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffLL)), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <typename F> __device__ __forceinline__ F grid_inf();
template <> __device__ __forceinline__ float grid_inf<float>() { return __builtin_inff(); }
template <> __device__ __forceinline__ double grid_inf<double>() { return __builtin_inf(); }

__device__ __forceinline__ int floor_to_int(float v) { return __float2int_rd(v); }
__device__ __forceinline__ int floor_to_int(double v) { return __double2int_rd(v); }

template <typename F>
__device__ __forceinline__ void grid_cell(const PatchGrid<F> &g, F x, F y, F z, int &cx, int &cy, int &cz) {
    cx = floor_to_int((x - g.minx) * g.inv_hx);
    cy = floor_to_int((y - g.miny) * g.inv_h);
    cz = floor_to_int((z - g.minz) * g.inv_hz);
}

// Squared distance: float32 with the fused chain fma(dz,dz, fma(dy,dy, dx*dx)); float64 with separately rounded
// multiply/add in x, y, z order, i.e. the bits Eigen's (a - b).squaredNorm() produces on x86-64 without FMA
// contraction (what the oracle and Open3D's CPU path evaluate).
__device__ __forceinline__ float grid_d2(float dx, float dy, float dz) {
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}
__device__ __forceinline__ double grid_d2(double dx, double dy, double dz) {
#pragma clang fp contract(off)
    double t = dx * dx;
    t = t + dy * dy;
    t = t + dz * dz;
    return t;
}

// ---- best-candidate record: lexicographic (d2, original index), plus the runner-up distance ---------------
// A grid point carries one 32-bit tag {original index : 16 | slot in tl : 16}; comparing tags compares original
// indices (the slot rides along for free and tells where the winner's coordinates are).  `second` is the
// smallest d2 offered that did NOT end up as the best: a lower bound on the distance of every other scanned
// target, which is what the nearest-neighbour certificates of icp.hip are made of (invariant: best <= second).
// float32: one 64-bit key {bits(d2) : tag}; d2 >= 0, so the IEEE bit pattern orders like the value.
constexpr unsigned int GRID_NO_TAG = 0xffffffffu;
__device__ __forceinline__ unsigned int grid_tag(int id, int slot) { return ((unsigned int)id << 16) | (unsigned int)slot; }

template <typename F> struct Best;
template <> struct Best<float> {
    unsigned long long key;
    float second;
    __device__ __forceinline__ void init(float bound2) {
        key = ((unsigned long long)__float_as_uint(bound2) << 32) | GRID_NO_TAG;
        second = __builtin_inff();
    }
    // branch free; a candidate at d2 = +inf (the dummy slot idle lanes read) changes nothing
    __device__ __forceinline__ void offer(float d, unsigned int tag) {
        const unsigned long long k = ((unsigned long long)__float_as_uint(d) << 32) | tag;
        second = __builtin_amdgcn_fmed3f(d, d2(), second);  // = min(second, max(d, best)) because best <= second
        key = k < key ? k : key;
    }
    __device__ __forceinline__ float d2() const { return __uint_as_float((unsigned int)(key >> 32)); }
    __device__ __forceinline__ unsigned int tag() const { return (unsigned int)(key & 0xffffffffULL); }
    __device__ __forceinline__ bool found() const { return tag() != GRID_NO_TAG; }
    __device__ __forceinline__ int id() const { return (int)(tag() >> 16); }
    __device__ __forceinline__ int slot() const { return (int)(tag() & 0xffffu); }
};
template <> struct Best<double> {
    double d, second;
    unsigned int t;
    __device__ __forceinline__ void init(double bound2) { d = bound2; t = GRID_NO_TAG; second = __builtin_inf(); }
    __device__ __forceinline__ void offer(double dd, unsigned int tag) {
        // branch free on purpose (the compiler turns the short-circuit form into two nested divergent branches per
        // candidate), and raw v_min / v_max: fmin / fmax would canonicalise both operands first
        const bool better = (dd < d) | ((dd == d) & (tag < t));
        double hi, lo, s2;
        asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(dd), "v"(d));  // the loser of the two is the larger distance (or an equal one)
        asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(dd), "v"(d));
        asm("v_min_f64 %0, %1, %2" : "=v"(s2) : "v"(second), "v"(hi));
        second = s2;
        d = lo;
        t = better ? tag : t;
    }
    __device__ __forceinline__ double d2() const { return d; }
    __device__ __forceinline__ unsigned int tag() const { return t; }
    __device__ __forceinline__ bool found() const { return t != GRID_NO_TAG; }
    __device__ __forceinline__ int id() const { return (int)(t >> 16); }
    __device__ __forceinline__ int slot() const { return (int)(t & 0xffffu); }
};

// ---- build ---------------------------------------------------------------------------------------------
// Called by all NT threads of the workgroup.  tg: packed float32 [nt][3] of the patch in global memory;
// (ox, oy, oz): patch origin; r: search radius (> 0); cell_cap: capacity of E in cells (E holds cell_cap + 2
// uint16, 4-byte aligned); red: NT/64 * 8 values of F scratch in LDS.  On return (after a barrier) tl and E
// are complete and g describes the grid.  1 <= nt <= 65534; tl holds nt + 1 records.
template <typename F, int NT>
__device__ __forceinline__ void grid_build(const float *__restrict__ tg, int nt, float ox, float oy, float oz, F r,
                                           int cell_cap, GridPt<F> *__restrict__ tl, unsigned short *__restrict__ E,
                                           F *__restrict__ red, PatchGrid<F> &g, int subdiv = 4, F dens = (F)4,
                                           int xsub = 4, bool allow_columns = true) {
    constexpr int NW = NT / 64;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // 1. bounding box of the patch (origin-relative)
    F mn[3] = {grid_inf<F>(), grid_inf<F>(), grid_inf<F>()}, mx[3] = {-grid_inf<F>(), -grid_inf<F>(), -grid_inf<F>()};
    for (int j = tid; j < nt; j += NT) {
        const F x = (F)tg[3 * j] - (F)ox, y = (F)tg[3 * j + 1] - (F)oy, z = (F)tg[3 * j + 2] - (F)oz;
        mn[0] = x < mn[0] ? x : mn[0]; mn[1] = y < mn[1] ? y : mn[1]; mn[2] = z < mn[2] ? z : mn[2];
        mx[0] = x > mx[0] ? x : mx[0]; mx[1] = y > mx[1] ? y : mx[1]; mx[2] = z > mx[2] ? z : mx[2];
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const F a = __shfl_xor(mn[d], m, 64), b = __shfl_xor(mx[d], m, 64);
            mn[d] = a < mn[d] ? a : mn[d];
            mx[d] = b > mx[d] ? b : mx[d];
        }
    }
    if (NW > 1) {
        if (lane == 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) { red[wave * 8 + d] = mn[d]; red[wave * 8 + 3 + d] = mx[d]; }
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NW; ++w) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const F a = red[w * 8 + d], b = red[w * 8 + 3 + d];
                mn[d] = a < mn[d] ? a : mn[d];
                mx[d] = b > mx[d] ? b : mx[d];
            }
        }
    }

    // 2. cell edge: r * (1 + 2^-7) / subdiv (patches that are dense relative to the radius get cells finer than the
    //    radius; the stencil then spans wmax = ceil(r (1 + 2^-7) / h) cells per side), enlarged until the grid fits
    //    the cell table (uniform across the workgroup).  NaN / inf coordinates degrade to a single cell (every
    //    query then scans the whole patch).
    const F ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
    const F rpad = r * (F)1.0078125;
    // a flat patch (terrain) gets one layer of cells: columns, unbounded in z
    const bool columns = allow_columns && ez <= (F)0.75 * (ex > ey ? ex : ey);
    auto cell_count = [&](F ih) {
        return (floor(ex * ih) + (F)1) * (floor(ey * ih) + (F)1) * (columns ? (F)1 : floor(ez * ih) + (F)1);
    };
    F h = rpad;
    if (subdiv > 1) {
        // subdivide only patches that are dense relative to the radius: the finest of r/1 .. r/subdiv that still leaves
        // `dens` points per cell of the bounding box on average (layered grids: surfaces leave most of the box empty, so
        // occupied cells hold several times that)
        int sd = 1;
        for (int c = 2; c <= subdiv; ++c) {
            if (cell_count((F)c / rpad) * dens <= (F)nt) sd = c;
        }
        h = rpad / (F)sd;
    }
    int nx = 1, ny = 1, nz = 1;
    const bool finite = (ex >= (F)0) && (ey >= (F)0) && (ez >= (F)0) && (ex + ey + ez < (F)1e30);
    if (finite) {
        for (int it = 0; it < 64; ++it) {
            const F ih = (F)1 / h;
            const F cells = cell_count(ih);
            if (cells <= (F)cell_cap) { nx = (int)(ex * ih) + 1; ny = (int)(ey * ih) + 1; nz = columns ? 1 : (int)(ez * ih) + 1; break; }
            if (it == 63) { h = (ex + ey + ez) * (F)2 + r; break; }  // gives a single cell
            F f = columns ? sqrt(cells / (F)cell_cap) : cbrt(cells / (F)cell_cap);  // exact for a volume; a surface needs a few rounds
            h *= f < (F)1.05 ? (F)1.05 : f;
        }
    }
    // cells along x: the finest of h / xsub, h / (xsub / 2) .. h that the table still holds
    int xs = 1;
    if (finite && nx * ny * nz > 0) {
        for (int c = xsub; c > 1; c >>= 1) {
            const F nxs = floor(ex * ((F)c / h)) + (F)1;
            if (nxs * (F)(ny * nz) <= (F)cell_cap) { xs = c; break; }
        }
    }
    g.minx = finite ? mn[0] : (F)0; g.miny = finite ? mn[1] : (F)0; g.minz = finite ? mn[2] : (F)0;
    g.ox = (F)ox; g.oy = (F)oy; g.oz = (F)oz;
    g.h = h; g.inv_h = (F)1 / h;
    g.inv_hx = (F)xs / h;
    g.inv_hz = (finite && nz > 1) ? g.inv_h : (F)0;
    if (finite && xs > 1) nx = (int)(ex * g.inv_hx) + 1;
    g.nx = nx; g.ny = ny; g.nz = nz;
    g.xs = xs;
    {
        const F wf = ceil(rpad / h - (F)1e-4);
        g.wmax = wf < (F)1 ? 1 : (wf > (F)4096 ? 4096 : (int)wf);
    }
    // every thread computed the same grid: tell the compiler, so that it lives in scalar registers
    g.minx = grid_uniform(g.minx); g.miny = grid_uniform(g.miny); g.minz = grid_uniform(g.minz);
    g.h = grid_uniform(g.h); g.inv_h = grid_uniform(g.inv_h);
    g.inv_hx = grid_uniform(g.inv_hx); g.inv_hz = grid_uniform(g.inv_hz);
    g.nx = __builtin_amdgcn_readfirstlane(g.nx); g.ny = __builtin_amdgcn_readfirstlane(g.ny);
    g.nz = __builtin_amdgcn_readfirstlane(g.nz); g.wmax = __builtin_amdgcn_readfirstlane(g.wmax);
    g.xs = __builtin_amdgcn_readfirstlane(g.xs);
    const int ncell = nx * ny * nz;

    // 3. histogram: E[c + 1] += 1 through 32-bit LDS atomics on the packed uint16 pairs
    unsigned int *Ew = reinterpret_cast<unsigned int *>(E);
    for (int i = tid; i < (ncell + 3) / 2; i += NT) Ew[i] = 0u;
    __syncthreads();
    for (int j = tid; j < nt; j += NT) {
        const F x = (F)tg[3 * j] - (F)ox, y = (F)tg[3 * j + 1] - (F)oy, z = (F)tg[3 * j + 2] - (F)oz;
        int cx, cy, cz;
        grid_cell(g, x, y, z, cx, cy, cz);
        cx = cx < 0 ? 0 : (cx >= nx ? nx - 1 : cx);
        cy = cy < 0 ? 0 : (cy >= ny ? ny - 1 : cy);
        cz = cz < 0 ? 0 : (cz >= nz ? nz - 1 : cz);
        const int e = (cz * ny + cy) * nx + cx + 1;
        atomicAdd(&Ew[e >> 1], (e & 1) ? 0x10000u : 1u);
    }
    __syncthreads();

    // 4. exclusive prefix over the counts: E[c + 1] <- number of points in cells < c  (E[0] stays 0)
    {
        const int chunk = (ncell + NT - 1) / NT;
        const int c0 = tid * chunk, c1 = (c0 + chunk < ncell) ? c0 + chunk : ncell;
        int sum = 0;
        for (int c = c0; c < c1; ++c) sum += (int)E[c + 1];
        // inclusive scan of the per-thread sums inside the wave (Hillis-Steele on shuffles; once per patch)
        int inc = sum;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const int v = __shfl_up(inc, m, 64);
            if (lane >= m) inc += v;
        }
        int *wsum = reinterpret_cast<int *>(red);
        if (NW > 1) {
            __syncthreads();  // `red` was read above by every thread
            if (lane == 63) wsum[wave] = inc;
            __syncthreads();
        }
        int base = inc - sum;
        if (NW > 1) {
#pragma unroll
            for (int w = 0; w < NW; ++w) base += (w < wave) ? wsum[w] : 0;
        }
        for (int c = c0; c < c1; ++c) {
            const int cnt = (int)E[c + 1];
            E[c + 1] = (unsigned short)base;
            base += cnt;
        }
    }
    __syncthreads();

    // 5. scatter: the slot E[c + 1] is the running cursor of cell c; when all points are placed it equals the
    //    start of cell c + 1, which is exactly the prefix table the queries read.
    for (int j = tid; j < nt; j += NT) {
        const F rx = (F)tg[3 * j] - (F)ox, ry = (F)tg[3 * j + 1] - (F)oy, rz = (F)tg[3 * j + 2] - (F)oz;
        GridPt<F> q;
        if (sizeof(F) == 4) { q.x = rx; q.y = ry; q.z = rz; }
        else { q.x = tg[3 * j]; q.y = tg[3 * j + 1]; q.z = tg[3 * j + 2]; }
        int cx, cy, cz;
        grid_cell(g, rx, ry, rz, cx, cy, cz);
        cx = cx < 0 ? 0 : (cx >= nx ? nx - 1 : cx);
        cy = cy < 0 ? 0 : (cy >= ny ? ny - 1 : cy);
        cz = cz < 0 ? 0 : (cz >= nz ? nz - 1 : cz);
        const int e = (cz * ny + cy) * nx + cx + 1;
        const unsigned int old = atomicAdd(&Ew[e >> 1], (e & 1) ? 0x10000u : 1u);
        const int pos = (int)((e & 1) ? (old >> 16) : (old & 0xffffu));
        q.tag = grid_tag(j, pos);
        tl[pos] = q;
    }
    if (tid == 0) {  // the dummy: farther than anything, never a winner
        GridPt<F> q;
        q.x = q.y = q.z = 1e30f;  // (squared: +inf in float32, 1e60 in float64 -- beyond any bound either way)
        q.tag = GRID_NO_TAG;
        tl[nt] = q;
    }
    __syncthreads();
}

// ---- query ---------------------------------------------------------------------------------------------
// Exact nearest target of (px, py, pz) among those with d2 < r2: every lane of the wave owns one query (lanes
// with valid == false own none) and ALL lanes of the wave must make the call together.
//
// `best` comes in init(b2)'d with an upper bound b2 <= (cell edge)^2 of the squared distance of interest (the
// correspondence radius, or last iteration's correspondence re-measured plus a margin); everything with
// d2 <= b2 is scanned, and the bound prunes the stencil BEFORE anything is scanned:
//   * the x-range of every row is narrowed to the cells within sqrt(best) of the query,
//   * rows (y, z) whose distance to the query exceeds the bound are dropped.
// The surviving non-empty rows are written as packed {start, end} pairs to a per-lane list in LDS
// (rl[k * NT + tid]: consecutive lanes -> consecutive banks) and then scanned by ONE flat, fully predicated
// loop: a single backward branch on a wave-uniform condition, next candidate and next row descriptor fetched
// one step ahead.  Both bounds are inclusive with slack for the rounding of the cell arithmetic, so exact
// ties on d2 with a smaller index are still seen.
constexpr int GRID_ROWS = 9;

template <typename F> __device__ __forceinline__ F grid_sqrt(F v);
template <> __device__ __forceinline__ float grid_sqrt<float>(float v) { return __builtin_amdgcn_sqrtf(v); }  // 1 ulp; the bound carries slack
// (float64 mode: every use is a bound that carries a 1e-6 relative allowance, so the float32 instruction -- good to
//  ~2e-7 after the conversion -- replaces the ~25-instruction IEEE double sequence)
template <> __device__ __forceinline__ double grid_sqrt<double>(double v) { return (double)__builtin_amdgcn_sqrtf((float)v); }

// Flat scan of the first `cnt` entries of this lane's row list; idle lanes sit on the dummy slot.  Returns the number
// of candidates this lane evaluated (profiling builds only use it).  `active`: lanes that take part (the others sit on the dummy).
template <typename F, typename FQ, int NT, class Measure>
__device__ __forceinline__ int grid_walk_rows(const GridPt<F> *__restrict__ tl, int dummy, const unsigned int *__restrict__ rl, int cnt,
                                              bool active, Best<FQ> &best, Measure &&measure) {
    const int tid = (int)threadIdx.x;
    int n_steps = 0;
    cnt = active ? cnt : 0;
    unsigned int cur = active ? rl[tid] : 0u;
    int j = (int)(cur & 0xffffu), e = (int)(cur >> 16);
    int k = cnt < 1 ? cnt : 1;
    // (a lane that sits the walk out must see the END of a list, not its own first row: entry `cnt` is the sentinel only for the lanes
    //  whose count was not overridden)
    unsigned int nxt = active ? rl[k * NT + tid] : 0u;
    GridPt<F> q = tl[j < e ? j : dummy];
    while (__any(j < e)) {
        // next position first, so that its loads are in flight while the current candidate is evaluated
        int jn = j + 1;
        const bool roll = jn >= e;
        jn = roll ? (int)(nxt & 0xffffu) : jn;
        const int en = roll ? (int)(nxt >> 16) : e;
        k = roll ? (k + 1 > cnt ? cnt : k + 1) : k;
        const GridPt<F> qn = tl[jn < en ? jn : dummy];
        const unsigned int nn = active ? rl[k * NT + tid] : 0u;
        // (float32: keep both requests up here -- left alone, the scheduler sinks them to the top of the next trip, a
        //  few instructions before their results are needed; +5 %.  float64: the longer evaluation hides them anyway
        //  and the pinned order costs 6 %, measured.)
        if (sizeof(FQ) == 4) __builtin_amdgcn_sched_barrier(0);
        // current candidate
        best.offer(measure(q), q.tag);
#ifdef F4L_ICP_PROF
        n_steps += j < e ? 1 : 0;
#endif
        q = qn; j = jn; e = en;
        nxt = roll ? nn : nxt;
    }
    return n_steps;
}
// float32 records, float32 arithmetic
template <typename F, int NT>
__device__ __forceinline__ int grid_scan_rows(const PatchGrid<float> &g, const GridPt<float> *__restrict__ tl, int dummy,
                                              const unsigned int *__restrict__ rl, int cnt, float px, float py, float pz, Best<float> &best) {
    return grid_walk_rows<float, float, NT>(tl, dummy, rl, cnt, true, best,
                                            [&](const GridPt<float> &q) { return grid_d2(px - q.x, py - q.y, pz - q.z); });
}
// float64 search (the parity mode).  The product walks the list in double (`exact` below).  -DF4L_GRID_PRESCAN builds the variant
// VERDICT r4 asked to have measured (item 2c, "float32 scan with a float64 guard band"): 18.50 against 18.59 ms at C4, +2 % at C3,
// but -9 % on a 1 M-point tile (C2: the 128-VGPR shapes spill what the second walk adds) -- profiles/r5_icp_occupancy_and_registers.log
// section 7; not enabled.  How it works: the list is walked in FLOAT32 first.  The query is split into its float32 image and
// a float32 remainder, Q = Qf + Qr to 2^-48 of itself; a record holds float32 coordinates, so (Qf - x) is one rounding of an exact
// difference (exact outright where it cancels) and (Qf - x) + Qr is dx to 1.2e-7 of itself whatever the magnitude of the coordinates
// -- d2 to 4e-7.  The walk keeps what the float64 walk keeps, in float32: the best (d2, tag) below the bound and the runner-up
// distance (Best<float>: the incoming bound counts as a loser once something beats it).  If the runner-up lies beyond the best by
// more than both errors can explain (a factor 1 + 4e-6), the float32 winner IS the minimiser of (exact d2, index) and beats the
// bound: its exact d2 is measured once, in double, the way the exact walk measures it; `second`, a LOWER bound on every other
// scanned target (what the certificates of icp.hip need of it), becomes the float32 runner-up less its error.  Lanes whose two best
// are closer than that -- duplicates, lattices, a candidate on the bound -- walk their list again in double: the answers are those
// of the exact walk in every case, only `second` may be lower by 2e-6 of itself.
template <typename F, int NT>
__device__ __forceinline__ int grid_scan_rows(const PatchGrid<double> &g, const GridPt<double> *__restrict__ tl, int dummy,
                                              const unsigned int *__restrict__ rl, int cnt, double px, double py, double pz, Best<double> &best) {
    const double Qx = grid_query(px, g.ox), Qy = grid_query(py, g.oy), Qz = grid_query(pz, g.oz);
    auto exact = [&](const GridPt<double> &q) { return grid_d2(Qx - (double)q.x, Qy - (double)q.y, Qz - (double)q.z); };
#ifndef F4L_GRID_PRESCAN
    return grid_walk_rows<double, double, NT>(tl, dummy, rl, cnt, true, best, exact);
#else
    const float fx = (float)Qx, fy = (float)Qy, fz = (float)Qz;
    const float rx = (float)(Qx - (double)fx), ry = (float)(Qy - (double)fy), rz = (float)(Qz - (double)fz);
    Best<float> bf;
    {
        const double b0 = best.d2() * 1.000002;  // (rounded up below: nothing at or under the bound is lost)
        bf.init(b0 < 3.0e38 ? __double2float_ru(b0) : __builtin_inff());
    }
    int steps = grid_walk_rows<double, float, NT>(tl, dummy, rl, cnt, true, bf, [&](const GridPt<double> &q) {
        const float dx = (fx - q.x) + rx, dy = (fy - q.y) + ry, dz = (fz - q.z) + rz;
        return grid_d2(dx, dy, dz);
    });
    const bool found = bf.found();
    const bool unclear = found && !(bf.second > bf.d2() * 1.000004f + 1e-37f);
    if (__any(unclear)) steps += grid_walk_rows<double, double, NT>(tl, dummy, rl, cnt, unclear, best, exact);
    if (!unclear) {
        const double lower = (double)bf.second * (1.0 - 2e-6);
        if (found) {
            const GridPt<double> w = tl[bf.slot()];
            best.d = exact(w);
            best.t = bf.tag();
        }
        best.second = lower < best.second ? lower : best.second;
    }
    return steps;
#endif
}

// The rare wide search of grid_nn (points without a previous correspondence on a grid finer than the radius): the
// (2 W + 1)^2 rows of the stencil, 9 at a time, every lane pruning with its own bound.  Written with rolled loops and
// a single scan site: it must not cost the common path registers or instruction-cache footprint.
template <typename F, int NT>
__device__ __forceinline__ Best<F> grid_nn_wide(const PatchGrid<F> &g, const GridPt<F> *__restrict__ tl, int dummy,
                                                       const unsigned short *__restrict__ E, unsigned int *__restrict__ rl,
                                                       int W, bool xok, int cy, int cz, int base0, int span, F fy, F fz,
                                                       F slack, F b2s, F px, F py, F pz, Best<F> best,
                                                       unsigned long long *prof = nullptr) {
    const int tid = (int)threadIdx.x;
    const int nxny = __mul24(g.nx, g.ny);
    // (dy, dz) walks the stencil row by row; both are uniform.  Layers and rows that no lane of the wave can use
    // (outside the grid -- surfaces leave most z layers empty -- or beyond every bound) cost one ballot, the others
    // are collected nine at a time and scanned.
    int dy = -W, dz = -W, r = 0, cnt = 0;
    for (;;) {
        if (dz <= W) {
            const int adz = dz < 0 ? -dz : dz;
            F ddz = dz == 0 ? (F)0 : (dz < 0 ? fz : g.h - fz) + (F)(adz - 1) * g.h - slack;
            ddz = ddz > (F)0 ? ddz : (F)0;
            const int z = cz + dz;
            const bool zk = xok && z >= 0 && z < g.nz && !(ddz * ddz > b2s);
            if (dy == -W && !__any(zk)) { ++dz; continue; }
            const int ady = dy < 0 ? -dy : dy;
            F ddy = dy == 0 ? (F)0 : (dy < 0 ? fy : g.h - fy) + (F)(ady - 1) * g.h - slack;
            ddy = ddy > (F)0 ? ddy : (F)0;
            const int y = cy + dy;
            const bool k = zk && y >= 0 && y < g.ny && !(ddy * ddy + ddz * ddz > b2s);
            if (__any(k)) {
                const int a = base0 + dy * g.nx + dz * nxny;
                const unsigned int s1 = E[k ? a : 0], e1 = E[k ? a + span : 0];
                rl[cnt * NT + tid] = s1 | (e1 << 16);
                cnt += s1 < e1 ? 1 : 0;
                ++r;
            }
            if (++dy > W) { dy = -W; ++dz; }
            if (r < GRID_ROWS && dz <= W) continue;
        }
        if (r > 0) {
            rl[cnt * NT + tid] = 0u;
#ifdef F4L_ICP_PROF
            const int st =
#endif
            grid_scan_rows<F, NT>(g, tl, dummy, rl, cnt, px, py, pz, best);
#ifdef F4L_ICP_PROF
            if (prof) {  // slot 13: scan rounds of the wide path, slot 4 + 16: its steps, slot 5 + 16: rows kept
                atomicAdd(&prof[20], (unsigned long long)st);
                atomicAdd(&prof[21], (unsigned long long)cnt);
                if (lane_id() == 0) { atomicAdd(&prof[13], 1ULL); atomicAdd(&prof[22], (unsigned long long)r); }
            }
#endif
            r = 0;
            cnt = 0;
        }
        if (dz > W) break;
    }
    return best;  // by value: a reference would pin the caller's record to memory on the common path too
}

// Largest search bound (a distance) that the 3 x 3 rows around a point's own cell are guaranteed to cover, slack of
// grid_nn included; <= 0 when rounding slack eats the whole cell.
template <typename F> __device__ __forceinline__ F grid_narrow_bound(const PatchGrid<F> &g, F px, F py, F pz) {
    const F slack = (F)1e-5 * g.h + (F)1e-6 * (fabs(px) + fabs(py) + fabs(pz) + fabs(g.minx) + fabs(g.miny) + fabs(g.minz));
    return (g.h - (F)2 * slack) * (F)0.9999;
}

// WIDE = false: the caller guarantees a bound below one cell edge (grid_narrow_bound), only the 3 x 3 path is emitted.
template <typename F, int NT, bool WIDE = true>
__device__ __forceinline__ void grid_nn(const PatchGrid<F> &g, const GridPt<F> *__restrict__ tl, int dummy,
                                        const unsigned short *__restrict__ E, unsigned int *__restrict__ rl, bool valid,
                                        F px, F py, F pz, Best<F> &best, unsigned long long *prof = nullptr) {
    const int tid = (int)threadIdx.x;
    int cx, cy, cz;
    grid_cell(g, px, py, pz, cx, cy, cz);
    // bound on the distance of anything that can still win, with slack for the rounding of cell coordinates
    const F slack = (F)1e-5 * g.h + (F)1e-6 * (fabs(px) + fabs(py) + fabs(pz) + fabs(g.minx) + fabs(g.miny) + fabs(g.minz));
    const F b2 = best.d2();
    const F bnd = grid_sqrt<F>(b2) * (F)1.00001 + slack;
    int x0 = floor_to_int((px - bnd - g.minx) * g.inv_hx), x1 = floor_to_int((px + bnd - g.minx) * g.inv_hx);
    const int wx = __mul24(g.wmax, g.xs);  // (x-cells are h / XS wide)
    x0 = x0 < cx - wx ? cx - wx : x0;  // never wider than the stencil the grid was built for
    x1 = x1 > cx + wx ? cx + wx : x1;
    x0 = x0 < 0 ? 0 : x0;
    x1 = x1 >= g.nx ? g.nx - 1 : x1;
    const bool xok = valid && x0 <= x1;
    // layers of cells this lane's bound reaches on either side of its own cell (y and z; x is the run itself)
    int W = 1;  // wave maximum (uniform); grids with cells no finer than the radius never need more than one layer
#ifndef F4L_GRID_NOWIDE
    if (WIDE && g.wmax > 1) {
        int wl = floor_to_int(bnd * g.inv_h) + 1;
        wl = valid ? (wl > g.wmax ? g.wmax : wl) : 0;
        if (__any(wl > 1)) {  // the widest stencil any lane of the wave needs (rows beyond a lane's own bound are pruned)
            W = g.wmax;
            while (W > 2 && !__any(wl >= W)) --W;
        }
    }
#endif
    const F fy = py - (g.miny + (F)cy * g.h), fz = pz - (g.minz + (F)cz * g.h);
    const F b2s = b2 * (F)1.00001;
    const int nxny = __mul24(g.nx, g.ny);
    const int base0 = __mul24(__mul24(cz, g.ny) + cy, g.nx) + x0;  // garbage when cy / cz are far outside: then no row is kept
    const int span = x1 - x0 + 1;
#ifdef F4L_ICP_PROF
    int n_steps = 0, cnt_total = 0;
    const unsigned long long pt0 = __builtin_readcyclecounter();
#endif

    if (W <= 1) {
        // ---- the common case: every bound of the wave fits the 3 x 3 layers around its cell
        F ylo = fy - slack, yhi = g.h - fy - slack, zlo = fz - slack, zhi = g.h - fz - slack;
        ylo = ylo > (F)0 ? ylo : (F)0; yhi = yhi > (F)0 ? yhi : (F)0;
        zlo = zlo > (F)0 ? zlo : (F)0; zhi = zhi > (F)0 ? zhi : (F)0;
        const F ysq[3] = {ylo * ylo, (F)0, yhi * yhi}, zsq[3] = {zlo * zlo, (F)0, zhi * zhi};
        const bool yin[3] = {cy >= 1 && cy <= g.ny, cy >= 0 && cy < g.ny, cy >= -1 && cy < g.ny - 1};
        const bool zin[3] = {cz >= 1 && cz <= g.nz, cz >= 0 && cz < g.nz, cz >= -1 && cz < g.nz - 1};
        // prefix entries of the 9 rows (centre row first, then the four face neighbours, then the corners), all
        // fetched before any is used; rows outside the grid or beyond the bound read entry 0 twice (empty)
        unsigned int s_[GRID_ROWS], e_[GRID_ROWS];
#pragma unroll
        for (int r = 0; r < GRID_ROWS; ++r) {
            constexpr int DY[GRID_ROWS] = {0, -1, 1, 0, 0, -1, 1, -1, 1};
            constexpr int DZ[GRID_ROWS] = {0, 0, 0, -1, 1, -1, -1, 1, 1};
            const bool k = xok && yin[DY[r] + 1] && zin[DZ[r] + 1] && !(ysq[DY[r] + 1] + zsq[DZ[r] + 1] > b2s);
            const int a = base0 + DY[r] * g.nx + DZ[r] * nxny;
            s_[r] = E[k ? a : 0];
            e_[r] = E[k ? a + span : 0];
        }
        // compact the non-empty rows into the per-lane list (branch free: every row is written at the cursor, the
        // cursor only moves past rows worth keeping), sentinel {0, 0} behind them
        int cnt = 0;
#pragma unroll
        for (int r = 0; r < GRID_ROWS; ++r) {
            rl[cnt * NT + tid] = s_[r] | (e_[r] << 16);
            cnt += s_[r] < e_[r] ? 1 : 0;
        }
        rl[cnt * NT + tid] = 0u;
#ifdef F4L_ICP_PROF
        cnt_total += cnt;
        n_steps +=
#endif
        grid_scan_rows<F, NT>(g, tl, dummy, rl, cnt, px, py, pz, best);
    } else {
#ifndef F4L_GRID_NOWIDE
#ifdef F4L_ICP_PROF
        if (prof) { atomicAdd(&prof[23], valid ? 1ULL : 0ULL); if (lane_id() == 0) { atomicAdd(&prof[24], 1ULL); atomicAdd(&prof[25], (unsigned long long)W); } }
        if (WIDE) best = grid_nn_wide<F, NT>(g, tl, dummy, E, rl, W, xok, cy, cz, base0, span, fy, fz, slack, b2s, px, py, pz, best, prof);
#else
        if (WIDE) best = grid_nn_wide<F, NT>(g, tl, dummy, E, rl, W, xok, cy, cz, base0, span, fy, fz, slack, b2s, px, py, pz, best);
#endif
#endif
    }
#ifdef F4L_ICP_PROF
    if (prof && lane_id() == 0) {
        const unsigned long long dt = __builtin_readcyclecounter() - pt0;
        atomicAdd(&prof[W <= 1 ? 27 : 28], dt);
        if (W > 1) atomicAdd(&prof[29], (unsigned long long)((2 * W + 1) * (2 * W + 1)));
        if (W >= 8) atomicAdd(&prof[30], 1ULL);
    }
    if (prof) {
        atomicAdd(&prof[6], (unsigned long long)n_steps);
        atomicAdd(&prof[7], (unsigned long long)(xok ? 9 : 0));
        atomicAdd(&prof[8], (unsigned long long)cnt_total);
        atomicAdd(&prof[9], valid ? 1ULL : 0ULL);
        int m = n_steps;
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) { const int o = __shfl_xor(m, sh, 64); m = o > m ? o : m; }
        if (lane_id() == 0) { atomicAdd(&prof[10], (unsigned long long)m); atomicAdd(&prof[11], 1ULL); }
    }
#endif
}

}  // namespace f4l
