// icp_rows.h -- per-patch point-to-point ICP for SMALL patches: several patches per wavefront, no workgroup barriers.
// (Included by icp.hip after IcpArgs and the solver helpers; same namespace.)
//
// The patches the reference's per-patch loop really meets are supervoxels of ~60 points
// (src/coarse_to_fine_matching_base.py:3254-3374 over `configs/landslide/fusion_brienz.yaml:25-26` tiles: thousands of them per
// tile; 16 766 patches, median 58 points, on the synthetic 1 M-point tile).  icp_kernel gives such a patch a wave of its own: the
// certify sweep is ONE batch, and the 17 sums' reduction and the solve -- 64 lanes doing one patch's arithmetic -- are half of the
// wave's life (profiles/r4_a_icp_phases_tile.log: solve 38 %, reduction 13 %, search 18 %, certify 12 %).  Here a patch takes
// LP = 16 or 32 lanes (one or two 16-lane DPP rows), a wave runs 64 / LP patches side by side, and every lane keeps up to four
// of its patch's source points:
//   * the points, their last correspondence, its certificate bound and the position of the last search live in REGISTERS
//     (four unrolled, independent chains per lane: the certify sweep needs no LDS but the target record it re-measures);
//   * the target patch sits in LDS as plain records in index order; a search is brute force over the <= 4 LP targets --
//     lane-per-query when many lanes of the wave ask at once (the first pass), cooperative otherwise: the LP lanes of a patch
//     split the targets of ONE queued point and merge their (best, runner-up) records through DPP;
//   * the row sums of row_sums_transposed ARE the patch's sums (two rows: added on read); every lane of a patch then runs the
//     solve on its patch's totals -- 64 / LP different solves for the issue slots of one;
//   * nothing ever waits for another wave: no barrier in the pass loop.
// Same arithmetic as icp_kernel in its per-point-certificate mode (exact (d2, index) minimiser, the certificates' triangle
// inequality, Open3D's loop and criteria); sums meet in another order, so results agree with it to rounding
// (test_icp_every_workgroup_shape_matches_oracle holds every shape to the oracle to 1e-9 m).
#pragma once

namespace f4l {

// All 16 lanes of a DPP row receive the row's merged record: quad stages, then two rotations.
template <int CTRL> __device__ __forceinline__ unsigned int dpp_u32(unsigned int v) {
    return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) { return dpp_mov<CTRL, 0xf>(v); }
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) { return dpp_mov<CTRL, 0xf>(v); }

// (best d2, its tag, runner-up d2) of two disjoint candidate sets -> of their union
template <typename F> struct Nn3 { F d; unsigned int t; F second; };
template <typename F> __device__ __forceinline__ Nn3<F> nn3_merge(const Nn3<F> &a, const Nn3<F> &b) {
    const bool a_wins = (a.d < b.d) | ((a.d == b.d) & (a.t <= b.t));
    Nn3<F> r;
    r.d = a_wins ? a.d : b.d;
    r.t = a_wins ? a.t : b.t;
    const F loser = a_wins ? b.d : a.d;
    F s = a.second < b.second ? a.second : b.second;
    r.second = loser < s ? loser : s;
    return r;
}
template <int CTRL> __device__ __forceinline__ Nn3<double> nn3_dpp_d(const Nn3<double> &v) {
    Nn3<double> o; o.d = dpp_f64<CTRL>(v.d); o.t = dpp_u32<CTRL>(v.t); o.second = dpp_f64<CTRL>(v.second); return o;
}
template <int CTRL> __device__ __forceinline__ Nn3<float> nn3_dpp_f(const Nn3<float> &v) {
    Nn3<float> o; o.d = dpp_f32<CTRL>(v.d); o.t = dpp_u32<CTRL>(v.t); o.second = dpp_f32<CTRL>(v.second); return o;
}
__device__ __forceinline__ Nn3<double> nn3_shfl16(const Nn3<double> &v) {
    Nn3<double> o; o.d = __shfl_xor(v.d, 16, 64); o.t = (unsigned int)__shfl_xor((int)v.t, 16, 64); o.second = __shfl_xor(v.second, 16, 64); return o;
}
__device__ __forceinline__ Nn3<float> nn3_shfl16(const Nn3<float> &v) {
    Nn3<float> o; o.d = __shfl_xor(v.d, 16, 64); o.t = (unsigned int)__shfl_xor((int)v.t, 16, 64); o.second = __shfl_xor(v.second, 16, 64); return o;
}
// merged over the LP lanes of a patch (every lane ends with the result); the lanes' candidate sets are disjoint, and a lane that
// holds none comes with d = bound, tag = GRID_NO_TAG (larger than every real tag), second = +inf
template <int LP> __device__ __forceinline__ Nn3<double> nn3_allreduce(Nn3<double> v) {
    v = nn3_merge(v, nn3_dpp_d<0xB1>(v));   // quad_perm [1,0,3,2]
    v = nn3_merge(v, nn3_dpp_d<0x4E>(v));   // quad_perm [2,3,0,1]
    v = nn3_merge(v, nn3_dpp_d<0x124>(v));  // row_ror:4
    v = nn3_merge(v, nn3_dpp_d<0x128>(v));  // row_ror:8
    if (LP == 32) v = nn3_merge(v, nn3_shfl16(v));
    return v;
}
template <int LP> __device__ __forceinline__ Nn3<float> nn3_allreduce(Nn3<float> v) {
    v = nn3_merge(v, nn3_dpp_f<0xB1>(v));
    v = nn3_merge(v, nn3_dpp_f<0x4E>(v));
    v = nn3_merge(v, nn3_dpp_f<0x124>(v));
    v = nn3_merge(v, nn3_dpp_f<0x128>(v));
    if (LP == 32) v = nn3_merge(v, nn3_shfl16(v));
    return v;
}
// NOTE on the rotations: after the two quad stages every lane holds its quad's record; rotating by 4 merges quad q with quad
// q + 1, rotating THAT by 8 merges pairs (q, q + 1) with (q + 2, q + 3): all four quads.  The merge is idempotent on the runner-up
// only for DISJOINT sets, and (q, q+1) against (q+2, q+3) are disjoint, as are q and q + 1.

// Sum over the LP lanes of a patch, every lane receives it (prologue only: a few values per patch).
template <int LP> __device__ __forceinline__ double group_sum(double v) {
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x124>(v);
    v += dpp_f64<0x128>(v);
    if (LP == 32) v += __shfl_xor(v, 16, 64);
    return v;
}

constexpr int ROWS_PPL = 4;  // source points per lane

// The rotation of an Umeyama step by the Jacobi SVD (rank-deficient or badly misaligned sums: rot_newton declined).  Rare, and
// kept out of line: inlined, its registers would be the kernel's peak.
__device__ __noinline__ void rows_rot_svd(const double *sg9, double *Ru) {
    double U[9], V[9];
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    svd3_warm(sg9, I3, U, V);
    const double sgn = (det3(U) * det3(V) < 0.0) ? -1.0 : 1.0;
    mul_diag_bt(U, sgn, V, Ru);
}

// One Umeyama / Newton update from a patch's 17 totals: the statements of icp_kernel's solving wave for MODE = point-to-point,
// on per-lane values (every lane of a patch computes its patch's update).  T: the patch's running transform in LDS (12 doubles:
// Rc row major, then tc; origin relative), updated in place by every lane of the patch with the same values.
__device__ __forceinline__ void rows_solve(const double (&tot)[17], int ns, int pass, int max_iter, int fixed_iters, double rel_fitness,
                                           double rel_rmse, int debug, double &fitness, double &rmse, int &iters, bool &done,
                                           double *T) {
    const double m = tot[0];
    const double fit_new = m > 0.0 ? m / (double)ns : 0.0;
    const double rmse_new = m > 0.0 ? sqrt(tot[1] / m) : 0.0;
    bool fin = false;
    if (pass > 0) {
        iters = pass;
        if (!fixed_iters && fabs(fitness - fit_new) < rel_fitness && fabs(rmse - rmse_new) < rel_rmse) fin = true;
    }
    if (pass == max_iter) fin = true;
    fitness = fit_new;
    rmse = rmse_new;
    done = fin;
    if (fin || !(m > 0.0)) return;
    const double im = fast_rcp(m);
    double sg9[9];
    const double cm0 = tot[2] * im, cm1 = tot[3] * im, cm2 = tot[4] * im;
    const double cq0 = tot[5] * im, cq1 = tot[6] * im, cq2 = tot[7] * im;
    sg9[0] = tot[8] * im - cq0 * cm0; sg9[1] = tot[9] * im - cq0 * cm1; sg9[2] = tot[10] * im - cq0 * cm2;
    sg9[3] = tot[11] * im - cq1 * cm0; sg9[4] = tot[12] * im - cq1 * cm1; sg9[5] = tot[13] * im - cq1 * cm2;
    sg9[6] = tot[14] * im - cq2 * cm0; sg9[7] = tot[15] * im - cq2 * cm1; sg9[8] = tot[16] * im - cq2 * cm2;
    double Ru[9];
    if ((debug & 128) || !rot_newton(sg9, Ru)) rows_rot_svd(sg9, Ru);
    const double tu0 = cq0 - (Ru[0] * cm0 + Ru[1] * cm1 + Ru[2] * cm2);
    const double tu1 = cq1 - (Ru[3] * cm0 + Ru[4] * cm1 + Ru[5] * cm2);
    const double tu2 = cq2 - (Ru[6] * cm0 + Ru[7] * cm1 + Ru[8] * cm2);
    double Rc[9], Rn[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Rc[i] = T[i];
    const double c0 = T[9], c1 = T[10], c2 = T[11];
    mul3(Ru, Rc, Rn);
    const double tn0 = Ru[0] * c0 + Ru[1] * c1 + Ru[2] * c2 + tu0;
    const double tn1 = Ru[3] * c0 + Ru[4] * c1 + Ru[5] * c2 + tu1;
    const double tn2 = Ru[6] * c0 + Ru[7] * c1 + Ru[8] * c2 + tu2;
    __builtin_amdgcn_wave_barrier();  // every lane of the patch has read the old transform
#pragma unroll
    for (int i = 0; i < 9; ++i) T[i] = Rn[i];
    T[9] = tn0; T[10] = tn1; T[11] = tn2;
}

// LDS per wave: targets (64 / LP patches x 4 LP records of 16 B = 4 KB) + the rows' sums (4 rows x 17 doubles)
constexpr int ROWS_TGT_PER_WAVE = 256;
constexpr int ROWS_WAVES = 4;
constexpr int ROWS_LDS_PER_WAVE = ROWS_TGT_PER_WAVE * 16 + 4 * 17 * 8 + 4 * 12 * 8 + 32;  // + the patches' running transforms
constexpr int ROWS_LDS = ROWS_WAVES * ROWS_LDS_PER_WAVE;

#ifndef ROWS_WPE
#define ROWS_WPE 3
#endif
template <typename F, int LP>
__global__ __launch_bounds__(ROWS_WAVES * 64, ROWS_WPE) void icp_rows_kernel(IcpArgs a) {
    constexpr int PW = 64 / LP;        // patches per wave
    constexpr int TCAP = ROWS_PPL * LP;  // targets (and sources) a patch may hold
    __shared__ __attribute__((aligned(16))) unsigned char rows_smem[ROWS_LDS];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane / LP, lr = lane % LP;
    unsigned char *wbase = rows_smem + wave * ROWS_LDS_PER_WAVE;
    GridPt<F> *tl = reinterpret_cast<GridPt<F> *>(wbase) + g * TCAP;
    double *sums = reinterpret_cast<double *>(wbase + ROWS_TGT_PER_WAVE * 16);
    double *Tl = sums + 4 * 17 + g * 12;  // this patch's running transform (origin relative): Rc row major, tc

    // which patch this group of lanes works on
    const int64_t slot = ((int64_t)blockIdx.x * ROWS_WAVES + wave) * PW + g;
    int64_t p = -1;
    if (a.list) { if (slot < (int64_t)*a.list_cnt) p = a.list[slot]; }
    else if (slot < a.P) p = slot;
    // A patch beyond the kernel's capacity is never loaded (its targets would overwrite the next patch's LDS) and is flagged
    // iters = -3; icp_launch_host only sends patches icp_bin_patches measured against ROWS_PPL * LP, so this guards other callers.
    bool oversize = false;
    if (p >= 0) {
        oversize = a.src_off[p + 1] - a.src_off[p] > (int64_t)TCAP || a.tgt_off[p + 1] - a.tgt_off[p] > (int64_t)TCAP;
        if (oversize) {
            if (lr == 0 && a.iters_out) a.iters_out[p] = -3;
            p = -1;
        }
    }
    const bool have_patch = p >= 0;
    const int64_t pp = have_patch ? p : 0;
    const int64_t s0 = a.src_off[pp], t0 = a.tgt_off[pp];
    const int ns = have_patch ? (int)(a.src_off[pp + 1] - s0) : 0, nt = have_patch ? (int)(a.tgt_off[pp + 1] - t0) : 0;
    const float *__restrict__ sg = a.src + 3 * s0;
    const float *__restrict__ tg = a.tgt + 3 * t0;
    const bool skipped = have_patch && a.corr_off != nullptr && a.corr_off[pp + 1] - a.corr_off[pp] < a.min_corr;
    const bool active = have_patch && ns > 0 && a.r2 > 0.0 && !skipped;
    float ox = 0.f, oy = 0.f, oz = 0.f;
    if (nt > 0) { ox = tg[0]; oy = tg[1]; oz = tg[2]; }
    else if (ns > 0) { ox = sg[0]; oy = sg[1]; oz = sg[2]; }

    // targets -> LDS, in index order.  float64 search: the ORIGINAL float32 coordinates (the distance is formed in the caller's
    // frame, like Open3D's); float32 search: patch relative.
    for (int k = lr; k < nt; k += LP) {
        GridPt<F> q;
        if (sizeof(F) == 4) { q.x = tg[3 * k] - ox; q.y = tg[3 * k + 1] - oy; q.z = tg[3 * k + 2] - oz; }
        else { q.x = tg[3 * k]; q.y = tg[3 * k + 1]; q.z = tg[3 * k + 2]; }
        q.tag = (unsigned int)k;
        tl[k] = q;
    }
    // this lane's source points (original float32 values; made relative on use) and their certificate state
    float sx[ROWS_PPL], sy[ROWS_PPL], sz[ROWS_PPL];
    int prev[ROWS_PPL];          // target index of the last correspondence; -1: nothing within the search radius; -2: never searched
    float mabs[ROWS_PPL];        // distance every OTHER target kept at the last search (rounded down)
    float psx[ROWS_PPL], psy[ROWS_PPL], psz[ROWS_PPL];  // position at the last search
#pragma unroll
    for (int m = 0; m < ROWS_PPL; ++m) {
        const int j = lr + m * LP;
        const int jj = j < ns ? j : (ns > 0 ? ns - 1 : 0);
        sx[m] = ns > 0 ? sg[3 * jj] : 0.f; sy[m] = ns > 0 ? sg[3 * jj + 1] : 0.f; sz[m] = ns > 0 ? sg[3 * jj + 2] : 0.f;
        prev[m] = -2; mabs[m] = 0.f; psx[m] = psy[m] = psz[m] = 0.f;
    }

    // initial transform: the fused Kabsch of the patch's correspondences (scripts/weighted_svd.py:58-129; the arithmetic of
    // icp_kernel's prologue), or init_T, or the identity
    double Tk[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    const bool fused_init = a.corr_off != nullptr;
    if (fused_init) {
        const int64_t c0 = a.corr_off[pp];
        const int nc = (!have_patch || skipped) ? 0 : (int)(a.corr_off[pp + 1] - c0);
        const float *__restrict__ ks = a.corr_src + 3 * c0, *__restrict__ kr = a.corr_ref + 3 * c0;
        const float *__restrict__ kw = a.corr_w ? a.corr_w + c0 : nullptr;
        double s7[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int i = lr; i < nc; i += LP) {
            double wi = kw ? (double)kw[i] : 1.0;
            if (wi < a.kabsch_w_thresh) wi = 0.0;
            s7[0] += wi;
            s7[1] += wi * (double)ks[3 * i]; s7[2] += wi * (double)ks[3 * i + 1]; s7[3] += wi * (double)ks[3 * i + 2];
            s7[4] += wi * (double)kr[3 * i]; s7[5] += wi * (double)kr[3 * i + 1]; s7[6] += wi * (double)kr[3 * i + 2];
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) s7[i] = group_sum<LP>(s7[i]);
        const double inv = 1.0 / (s7[0] + a.kabsch_eps);  // eps stays in the denominator (weighted_svd.py:96)
        const double k0 = s7[1] * inv, k1 = s7[2] * inv, k2 = s7[3] * inv;
        const double l0 = s7[4] * inv, l1 = s7[5] * inv, l2 = s7[6] * inv;
        double h9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = lr; i < nc; i += LP) {
            double wi = kw ? (double)kw[i] : 1.0;
            if (wi < a.kabsch_w_thresh) wi = 0.0;
            wi *= inv;
            const double a0 = (double)ks[3 * i] - k0, a1 = (double)ks[3 * i + 1] - k1, a2 = (double)ks[3 * i + 2] - k2;
            const double b0 = wi * ((double)kr[3 * i] - l0), b1 = wi * ((double)kr[3 * i + 1] - l1),
                         b2 = wi * ((double)kr[3 * i + 2] - l2);
            h9[0] += a0 * b0; h9[1] += a0 * b1; h9[2] += a0 * b2;
            h9[3] += a1 * b0; h9[4] += a1 * b1; h9[5] += a1 * b2;
            h9[6] += a2 * b0; h9[7] += a2 * b1; h9[8] += a2 * b2;
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) h9[i] = group_sum<LP>(h9[i]);
        if (nc > 0) {
            double R[9];
            const double Ht[9] = {h9[0], h9[3], h9[6], h9[1], h9[4], h9[7], h9[2], h9[5], h9[8]};
            if (!rot_newton(Ht, R)) {
                double U[9], V[9];
                const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
                svd3_warm(h9, I3, U, V);
                const double dd = det3(V) * det3(U);
                mul_diag_bt(V, dd > 0.0 ? 1.0 : (dd < 0.0 ? -1.0 : 0.0), U, R);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) { Tk[4 * i] = R[3 * i]; Tk[4 * i + 1] = R[3 * i + 1]; Tk[4 * i + 2] = R[3 * i + 2]; }
            Tk[3] = l0 - (R[0] * k0 + R[1] * k1 + R[2] * k2);
            Tk[7] = l1 - (R[3] * k0 + R[4] * k1 + R[5] * k2);
            Tk[11] = l2 - (R[6] * k0 + R[7] * k1 + R[8] * k2);
            if (a.init_round_f32) {
#pragma unroll
                for (int i = 0; i < 12; ++i) Tk[i] = (double)(float)Tk[i];
            }
        }
    } else if (a.init_T && have_patch) {
        const double *T = a.init_T + 16 * pp;
#pragma unroll
        for (int i = 0; i < 12; ++i) Tk[i] = T[i];
    }
    // running transform in origin-relative coordinates, p' = Rc s' + tc, kept in LDS (every lane of the patch writes the same values)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        Tl[3 * i] = Tk[4 * i]; Tl[3 * i + 1] = Tk[4 * i + 1]; Tl[3 * i + 2] = Tk[4 * i + 2];
        Tl[9 + i] = Tk[4 * i] * (double)ox + Tk[4 * i + 1] * (double)oy + Tk[4 * i + 2] * (double)oz + Tk[4 * i + 3] -
                    (double)(i == 0 ? ox : (i == 1 ? oy : oz));
    }
    double fitness = 0.0, rmse = 0.0;
    int iters = 0;
    bool fin = !active;

    const F rF = (F)a.r, r2 = (F)a.r2;
    const F rs = rF * (F)1.0625, rs2 = rs * rs;  // search a little beyond the correspondence radius: "nothing within r" is certified too
    // the largest target count of the wave's patches: the trip count of the lane-per-query scan (wave uniform)
    int nt_max = nt;
#pragma unroll
    for (int sh = LP; sh < 64; sh <<= 1) { const int o = __shfl_xor(nt_max, sh, 64); nt_max = o > nt_max ? o : nt_max; }
    nt_max = __builtin_amdgcn_readfirstlane(nt_max);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the wave's own LDS writes (targets) before its reads
    __builtin_amdgcn_wave_barrier();

    const int n_pass = a.max_iter + 1;
    for (int pass = 0; pass < n_pass; ++pass) {
        if (__ballot(!fin) == 0ULL) break;  // every patch of the wave has finished
        F px[ROWS_PPL], py[ROWS_PPL], pz[ROWS_PPL];
        {
            const F R0 = (F)Tl[0], R1 = (F)Tl[1], R2 = (F)Tl[2], R3 = (F)Tl[3], R4 = (F)Tl[4], R5 = (F)Tl[5], R6 = (F)Tl[6],
                    R7 = (F)Tl[7], R8 = (F)Tl[8];
            const F t0f = (F)Tl[9], t1f = (F)Tl[10], t2f = (F)Tl[11];
#pragma unroll
            for (int m = 0; m < ROWS_PPL; ++m) {
                const F x = (F)sx[m] - (F)ox, y = (F)sy[m] - (F)oy, z = (F)sz[m] - (F)oz;
                px[m] = R0 * x + R1 * y + R2 * z + t0f;
                py[m] = R3 * x + R4 * y + R5 * z + t1f;
                pz[m] = R6 * x + R7 * y + R8 * z + t2f;
            }
        }
        bool need[ROWS_PPL];
        int hit_k[ROWS_PPL];  // the accepted correspondence of the pass (target index), -1: none; summed after the searches
        F hit_d[ROWS_PPL];
        // squared distance of a moved point to a target record, the bits the reference's arithmetic gives (patch_grid.h)
        auto dist2 = [&](F ppx, F ppy, F ppz, const GridPt<F> &q) {
            return grid_d2(grid_query(ppx, (F)ox) - grid_coord(q.x, ppx), grid_query(ppy, (F)oy) - grid_coord(q.y, ppy),
                           grid_query(ppz, (F)oz) - grid_coord(q.z, ppz));
        };

        // ---- certify: re-measure the last correspondence of each of this lane's points (four independent chains)
#pragma unroll
        for (int m = 0; m < ROWS_PPL; ++m) {
            const int j = lr + m * LP;
            const bool valid = j < ns && !fin;
            const int pv = prev[m];
            // how far the point has moved since its last search (float32: the stored position is float32 anyway), rounded up
            const float fx = (float)px[m], fy = (float)py[m], fz = (float)pz[m];
            const float mx = fx - psx[m], my = fy - psy[m], mz = fz - psz[m];
            float moved = __builtin_amdgcn_sqrtf(__builtin_fmaf(mz, mz, __builtin_fmaf(my, my, mx * mx))) * 1.000001f;
            moved += 2e-7f * (fabsf(fx) + fabsf(fy) + fabsf(fz));
            const float room = mabs[m] - moved;
            const GridPt<F> q = tl[pv >= 0 ? pv : 0];
            const F d = dist2(px[m], py[m], pz[m], q);
            bool cert = pv >= 0 ? (float)grid_sqrt<F>(d) * 1.000001f < room : (pv == -1 && room > (float)rF * 1.000001f);
            cert = cert && valid;
            const bool hit = cert && pv >= 0 && d < r2;
            hit_k[m] = hit ? pv : -1;
            hit_d[m] = d;
            if (a.corr_out && cert) a.corr_out[s0 + j] = hit ? pv : -1;
            need[m] = valid && !cert;
        }

        // ---- search the points whose certificate did not hold
#pragma unroll
        for (int m = 0; m < ROWS_PPL; ++m) {
            unsigned long long mask = __ballot(need[m]);
            if (mask == 0ULL) continue;
            const int j = lr + m * LP;
            F bd = rs2, bsecond = grid_inf<F>();
            unsigned int bt = GRID_NO_TAG;
            bool mine = false;  // this lane's point m has just been searched: (bd, bt, bsecond) is its result
            if (__builtin_popcountll(mask) >= 20) {
                // many at once (the first pass): every asking lane scans its patch's targets for its own point
                Best<F> best;
                best.init(rs2);
                const bool ask = need[m];
                for (int k = 0; k < nt_max; ++k) {
                    const bool in = ask && k < nt;
                    const GridPt<F> q = tl[in ? k : 0];
                    F d = dist2(px[m], py[m], pz[m], q);
                    d = in ? d : grid_inf<F>();
                    best.offer(d, in ? (unsigned int)k : GRID_NO_TAG);
                }
                mine = ask;
                bd = best.d2(); bt = best.tag(); bsecond = best.second;
            } else {
                // a few: one queued point per patch at a time, the patch's LP lanes split its targets
                while (mask != 0ULL) {
                    const unsigned int gm = (unsigned int)((mask >> (g * LP)) & (LP == 32 ? 0xffffffffULL : 0xffffULL));
                    const bool has = gm != 0u;
                    const int owner = g * LP + (has ? __builtin_ctz(gm) : 0);
                    const F qpx = __shfl(px[m], owner, 64), qpy = __shfl(py[m], owner, 64), qpz = __shfl(pz[m], owner, 64);
                    Nn3<F> v;
                    v.d = rs2; v.t = GRID_NO_TAG; v.second = grid_inf<F>();
#pragma unroll
                    for (int c = 0; c < ROWS_PPL; ++c) {
                        const int k = lr + c * LP;
                        const bool in = has && k < nt;
                        const GridPt<F> q = tl[in ? k : 0];
                        F d = dist2(qpx, qpy, qpz, q);
                        d = in ? d : grid_inf<F>();
                        const unsigned int kt = in ? (unsigned int)k : GRID_NO_TAG;
                        // (a candidate at or beyond the bound never becomes the best: it only feeds the runner-up)
                        const bool better = (d < v.d) | ((d == v.d) & (kt < v.t));
                        const F loser = better ? v.d : d;
                        v.second = loser < v.second ? loser : v.second;
                        v.d = better ? d : v.d;
                        v.t = better ? kt : v.t;
                    }
                    v = nn3_allreduce<LP>(v);
                    const bool own = has && lane == owner;
                    if (own) { mine = true; bd = v.d; bt = v.t; bsecond = v.second; }
                    mask &= ~__ballot(own);
                }
            }
            if (mine) {
                const F m2 = bsecond < rs2 ? bsecond : rs2;
                const bool found = bt != GRID_NO_TAG;
                prev[m] = found ? (int)bt : -1;
                mabs[m] = (float)grid_sqrt<F>(m2) * 0.999999f;
                psx[m] = (float)px[m]; psy[m] = (float)py[m]; psz[m] = (float)pz[m];
                const bool hit = found && bd < r2;  // SearchHybrid: d2 < r^2
                hit_k[m] = hit ? (int)bt : -1;
                hit_d[m] = bd;
                if (a.corr_out) a.corr_out[s0 + j] = hit ? (int)bt : -1;
            }
        }

        // ---- the accepted pairs of the pass -> the 17 sums (double, uncentred: the reference's arithmetic), then the patch's totals:
        //      the row sums of the transposing reduction ARE a patch's sums (two rows per patch: added on read)
        double tot[17];
        {
            double acc[17];
#pragma unroll
            for (int i = 0; i < 17; ++i) acc[i] = 0.0;
#pragma unroll
            for (int m = 0; m < ROWS_PPL; ++m) {
                const bool hit = hit_k[m] >= 0;
                const GridPt<F> q = tl[hit ? hit_k[m] : 0];
                double qx, qy, qz;
                if (sizeof(F) == 4) { qx = (double)q.x; qy = (double)q.y; qz = (double)q.z; }
                else { qx = (double)q.x - (double)ox; qy = (double)q.y - (double)oy; qz = (double)q.z - (double)oz; }
                // (branch free: a lane without a pair adds zeros -- selected, not multiplied: the record read for it may hold anything)
                const double dpx = hit ? (double)px[m] : 0.0, dpy = hit ? (double)py[m] : 0.0, dpz = hit ? (double)pz[m] : 0.0;
                qx = hit ? qx : 0.0; qy = hit ? qy : 0.0; qz = hit ? qz : 0.0;
                acc[0] += hit ? 1.0 : 0.0;
                acc[1] += hit ? (double)hit_d[m] : 0.0;
                acc[2] += dpx; acc[3] += dpy; acc[4] += dpz;
                acc[5] += qx; acc[6] += qy; acc[7] += qz;
                acc[8] += qx * dpx; acc[9] += qx * dpy; acc[10] += qx * dpz;
                acc[11] += qy * dpx; acc[12] += qy * dpy; acc[13] += qy * dpz;
                acc[14] += qz * dpx; acc[15] += qz * dpy; acc[16] += qz * dpz;
            }
            double xs[4], ys[1];
            row_sums_transposed<17, double>(acc, xs, ys);
            if ((lane & 12) == 0) {
                double *row = sums + (lane >> 4) * 17;
#pragma unroll
                for (int i = 0; i < 4; ++i) row[4 * i + (lane & 3)] = xs[i];
                if ((lane & 3) == 0) row[16] = ys[0];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const double *mine_rows = sums + (LP == 32 ? 2 * g : g) * 17;
#pragma unroll
            for (int i = 0; i < 17; ++i) tot[i] = LP == 32 ? mine_rows[i] + mine_rows[17 + i] : mine_rows[i];
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // (the next pass rewrites the rows)
        }
        // ---- every lane solves its patch (64 / LP different solves in the issue slots of one)
        if (!fin) {
            bool done = false;
            rows_solve(tot, ns, pass, a.max_iter, a.fixed_iters, a.rel_fitness, a.rel_rmse, a.debug, fitness, rmse, iters, done, Tl);
            fin = done;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the new transforms before the next pass reads them
    }

    // ---- results: back to the caller's coordinates, t = tc - Rc o + o
    const double o0 = ox, o1 = oy, o2 = oz;
    double Rc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Rc[i] = Tl[i];
    const double tr0 = Tl[9] - (Rc[0] * o0 + Rc[1] * o1 + Rc[2] * o2) + o0;
    const double tr1 = Tl[10] - (Rc[3] * o0 + Rc[4] * o1 + Rc[5] * o2) + o1;
    const double tr2 = Tl[11] - (Rc[6] * o0 + Rc[7] * o1 + Rc[8] * o2) + o2;
    if (have_patch && lr == 0) {
        double *T = a.T_out + 16 * p;
#pragma unroll
        for (int i = 0; i < 3; ++i) { T[4 * i] = Rc[3 * i]; T[4 * i + 1] = Rc[3 * i + 1]; T[4 * i + 2] = Rc[3 * i + 2]; }
        T[3] = tr0; T[7] = tr1; T[11] = tr2;
        T[12] = 0.0; T[13] = 0.0; T[14] = 0.0; T[15] = 1.0;
        if (a.fitness_out) a.fitness_out[p] = fitness;
        if (a.rmse_out) a.rmse_out[p] = rmse;
        if (a.iters_out) a.iters_out[p] = skipped ? -1 : iters;
    }
    if (have_patch && !active && a.corr_out)
        for (int i = lr; i < ns; i += LP) a.corr_out[s0 + i] = -1;
    if (have_patch && a.rows_out && !skipped) {
        // displacement rows [s, T s] (src/coarse_to_fine_matching_base.py:3371-3374,3408): the arithmetic of apply_transform_kernel
        const int64_t w0 = a.rows_off ? a.rows_off[p] : s0;
        const int nrow = a.rows_off ? (int)(a.rows_off[p + 1] - w0) : ns;
        const float *__restrict__ wg = a.rows_src ? a.rows_src + 3 * w0 : sg;
        float *__restrict__ out6 = a.rows_out + 6 * w0;
        for (int i = lr; i < nrow; i += LP) {
            const float xf = wg[3 * i], yf = wg[3 * i + 1], zf = wg[3 * i + 2];
            const double x = xf, y = yf, z = zf;
            float *o6 = out6 + 6 * i;
            o6[0] = xf; o6[1] = yf; o6[2] = zf;
            o6[3] = (float)(Rc[0] * x + Rc[1] * y + Rc[2] * z + tr0);
            o6[4] = (float)(Rc[3] * x + Rc[4] * y + Rc[5] * z + tr1);
            o6[5] = (float)(Rc[6] * x + Rc[7] * y + Rc[8] * z + tr2);
        }
    }
}

}  // namespace f4l
