// EXPERIMENT, NOT PART OF THE BUILD (round 6; measured and not kept -- see the note in csrc/icp.hip `launch_icp` and DESIGN.md).
// To try it again: copy next to icp.hip, include it after icp_rows.h, add `int lazy` to IcpArgs with the eligibility test, and launch
// icp_lazy_kernel<NW> before icp_kernel with IcpArgs::lazy = 1 (git history of round 6 has the wiring).
// icp_lazy.h -- point-to-point ICP of one patch pair per workgroup with LAZY correspondence sums (round 6; included by icp.hip).
//
// icp_kernel re-measures every source point in every pass: transform it, measure the distance to last pass's correspondent, certify
// that correspondent (or queue the point for a search), and add the pair's 17 products to the pass's sums.  After the first few
// passes nothing of that changes any more -- a patch's 20 fixed iterations (BASELINE.json's metric) spend three quarters of their
// passes re-deriving the same correspondence set from positions that moved by micrometres.  Here the sums are kept LAZILY:
//
//   moments   The pairs that currently count (certified or searched correspondences with d < r) are held as 18 sums in the patch's
//             ORIGINAL coordinates, taken about the source patch's centroid c -- n, sum x, sum q, sum q x^T, sum |x|^2, sum |q|^2
//             (x: source point - c, q: its target - c) -- and a pass obtains its Umeyama quantities under the current transform
//             p = R x + t' (t' = the centroid's image, which the kernel tracks anyway) from them in O(1):
//                 mean p = R (sum x) / n + t',   mean q = (sum q) / n + c,
//                 cov(q, p) = (sum q x^T) R^T / n - (sum q / n) (R sum x / n)^T,
//                 sum |p - q|^2 = sum |x|^2 + 2 s . R sum x + n |s|^2 - 2 <R, sum q x^T> - 2 s . sum q + sum |q|^2,   s = t' - c.
//             The last line cancels (patch extent^2 against residual^2): good to ~1e-12 of the rmse on scanned surfaces; a patch whose
//             rmse is so small that it would show (clouds that fit to micrometres) has the sum of squares measured point by point
//             instead, in the arithmetic icp_kernel uses (`direct` passes).
//   deadlines Every point carries a DEADLINE in units of the patch-wide motion bound B that icp_kernel already accumulates
//             (state[29]: |p_new - p_old| <= ||Ru - I||_F radius + |centroid step| per update, summed): while B stays below it, the
//             point's certificate (its correspondent is still the nearest target) AND its status (inside / outside the radius) are
//             guaranteed by the triangle inequality -- the point is not touched at all.  A point whose deadline has passed is
//             re-measured exactly like icp_kernel's phase 1 does (its own displacement since its last search, not the patch-wide
//             bound): still certified -> a new deadline from where it stands now; else it is searched, like there.
//   deltas    Only a point whose status or correspondent CHANGES touches the moments: minus its old pair, plus its new one.
//
// The correspondence set of every pass is the one icp_kernel finds (both skip only what is proven unchanged), so the sums agree to
// rounding (they are taken in another order and frame), and transforms, fitness and rmse to the tolerances of tests/test_gpu_parity.py
// (1e-9 m, equal iteration counts).  A quiet pass costs a scan of the deadlines, the solve and its barriers.
// Eligible: point-to-point, float64 search, targets in LDS, certificates with per-point positions (ns <= cert_cap, pp_cap);
// every other patch, mode or precision stays with icp_kernel (which skips the patches this kernel takes: IcpArgs::lazy).
#pragma once

namespace f4l {

constexpr int LAZY_NV = 19;  // n, sum x (3), sum q (3), sum q_a x_b (9), sum |x|^2, sum |q|^2; slot 18: a `direct` pass's sum of d^2 (not a moment)
constexpr int LAZY_NM = 18;  // the moments among them
constexpr int LAZY_STATE = 72;  // doubles of `state`: icp_kernel's 48 (33: the next pass is a `direct` one) + the moments at 48..65

// bytes of dynamic LDS the lazy kernel needs for a plan of icp_kernel's (same arrays, plus the deadlines and the wider state)
static inline size_t icp_lazy_lds_bytes(int nw, int tgt_cap, int cell_cap, int cert_cap, int src_cap, int pp_cap) {
    const int nt = nw * 64;
    const size_t sum_doubles = (size_t)((4 * nw * LAZY_NV + 1) & ~1);
    size_t b = (sum_doubles + LAZY_STATE) * sizeof(double) + 16;
    b += (size_t)(tgt_cap + 1) * sizeof(GridPt<double>);
    b += (size_t)((cert_cap + 3) & ~3) * 4;       // mabs
    b += (size_t)((cert_cap + 3) & ~3) * 4;       // deadlines
    b += (size_t)((pp_cap + 3) & ~3) * 3 * 4;     // ps
    b += (size_t)((src_cap + 3) & ~3) * 3 * 8;    // sl
    b += (size_t)(GRID_ROWS + 1) * nt * sizeof(unsigned int);
    b += ((size_t)cell_cap + 8) * 2;
    b += (size_t)((cert_cap + 7) & ~7) * 2;       // prev
    b += (size_t)nw * (size_t)((cert_cap + nt - 1) / nt) * 64 * 2 + 16;  // queue
    return (b + 15) & ~(size_t)15;
}


template <int NW>
__global__ __launch_bounds__(NW * 64, 3) void icp_lazy_kernel(IcpArgs a) {
    using F = double;
    constexpr int NV = LAZY_NV;
    constexpr int NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int SUM_ROWS = 4 * NW;
    constexpr int SCRATCH = (SUM_ROWS * NV + 1) & ~1;
    double *scratch = reinterpret_cast<double *>(smem_raw);
    double *state = scratch + SCRATCH;
    double *M = state + 48;  // the moments
    int *qcnt = reinterpret_cast<int *>(state + LAZY_STATE);
    GridPt<F> *tl = reinterpret_cast<GridPt<F> *>(qcnt + 4);
    float *mabs = reinterpret_cast<float *>(tl + a.tgt_cap + 1);
    float *dl = mabs + ((a.cert_cap + 3) & ~3);  // deadline (units of the motion bound); sign bit set: the point's pair is in the moments
    float *ps = dl + ((a.cert_cap + 3) & ~3);
    F *sl = reinterpret_cast<F *>(ps + 3 * ((a.pp_cap + 3) & ~3));
    unsigned int *rl = reinterpret_cast<unsigned int *>(sl + 3 * ((a.src_cap + 3) & ~3));
    unsigned short *E = reinterpret_cast<unsigned short *>(rl + (GRID_ROWS + 1) * NT);
    unsigned short *prev = E + a.cell_cap + 8;
    const int seg = ((a.cert_cap + NT - 1) / NT) * 64;
    unsigned short *queue = prev + ((a.cert_cap + 7) & ~7);

    int64_t p = blockIdx.x;
    if (a.list) {
        if ((int)blockIdx.x >= *a.list_cnt) return;
        p = a.list[blockIdx.x];
    }
    if (p >= a.P) return;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t s0 = a.src_off[p], t0 = a.tgt_off[p];
    const int ns = (int)(a.src_off[p + 1] - s0), nt = (int)(a.tgt_off[p + 1] - t0);
    if (!icp_lazy_eligible(a, ns, nt)) return;  // (icp_kernel's patch)
    const float *__restrict__ sg = a.src + 3 * s0;
    const float *__restrict__ tg = a.tgt + 3 * t0;
    const bool skipped = a.corr_off != nullptr && a.corr_off[p + 1] - a.corr_off[p] < a.min_corr;
    const bool active = a.r2 > 0.0 && !skipped;
    const bool src_in_lds = ns <= a.src_cap;
    const float ox = tg[0], oy = tg[1], oz = tg[2];  // per-patch origin: first target point

    const F rF = (F)a.r, r2 = (F)a.r2;
    const F rs = rF * (F)1.0625, rs2 = rs * rs;
    PatchGrid<F> g;
    g.minx = g.miny = g.minz = (F)0; g.h = (F)1; g.inv_h = (F)1; g.inv_hx = (F)1; g.inv_hz = (F)1; g.nx = g.ny = g.nz = 1; g.xs = 1; g.wmax = 1;
    double cs[3] = {0.0, 0.0, 0.0}, srad = 0.0;
    if (active) {
        grid_build<F, NT>(tg, nt, ox, oy, oz, rs, a.cell_cap, tl, E, reinterpret_cast<F *>(scratch), g, a.subdiv, (F)a.dens, a.xsub, true);
        double sum[3] = {0.0, 0.0, 0.0};
        for (int i = tid; i < ns; i += NT) {
            sum[0] += (double)((F)sg[3 * i] - (F)ox); sum[1] += (double)((F)sg[3 * i + 1] - (F)oy); sum[2] += (double)((F)sg[3 * i + 2] - (F)oz);
            prev[i] = 0xffffu; mabs[i] = 0.f; dl[i] = 0.f;
            if (src_in_lds) { sl[3 * i] = (F)sg[3 * i] - (F)ox; sl[3 * i + 1] = (F)sg[3 * i + 1] - (F)oy; sl[3 * i + 2] = (F)sg[3 * i + 2] - (F)oz; }
        }
        block_sum<3, NW>(sum, scratch);
        cs[0] = sum[0] / (double)ns; cs[1] = sum[1] / (double)ns; cs[2] = sum[2] / (double)ns;
        double mx2 = 0.0;
        for (int i = tid; i < ns; i += NT) {
            const double dx = (double)((F)sg[3 * i] - (F)ox) - cs[0], dy = (double)((F)sg[3 * i + 1] - (F)oy) - cs[1],
                         dz = (double)((F)sg[3 * i + 2] - (F)oz) - cs[2];
            const double d2 = dx * dx + dy * dy + dz * dz;
            mx2 = d2 > mx2 ? d2 : mx2;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { const double o = __shfl_xor(mx2, m, 64); mx2 = o > mx2 ? o : mx2; }
        if (NW > 1) {
            __syncthreads();
            if (lane == 0) scratch[wave] = mx2;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < NW; ++w) mx2 = scratch[w] > mx2 ? scratch[w] : mx2;
        }
        srad = sqrt(mx2) * (1.0 + 1e-6);
    }
    __syncthreads();
    const F mcell = g.h * (F)(g.wmax > 1 ? a.mu_cell_fine : a.mu_cell);
    const F mu = rF * (F)a.mu_frac < mcell ? rF * (F)a.mu_frac : mcell;

    // fused initialisation: weighted Kabsch of this patch's correspondences (as icp_kernel's)
    double Tk[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    const bool fused_init = a.corr_off != nullptr;
    if (fused_init) {
        const int64_t c0 = a.corr_off[p];
        const int nc = skipped ? 0 : (int)(a.corr_off[p + 1] - c0);
        const float *__restrict__ ks = a.corr_src + 3 * c0, *__restrict__ kr = a.corr_ref + 3 * c0;
        const float *__restrict__ kw = a.corr_w ? a.corr_w + c0 : nullptr;
        double s7[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int i = tid; i < nc; i += NT) {
            double wi = kw ? (double)kw[i] : 1.0;
            if (wi < a.kabsch_w_thresh) wi = 0.0;
            s7[0] += wi;
            s7[1] += wi * (double)ks[3 * i]; s7[2] += wi * (double)ks[3 * i + 1]; s7[3] += wi * (double)ks[3 * i + 2];
            s7[4] += wi * (double)kr[3 * i]; s7[5] += wi * (double)kr[3 * i + 1]; s7[6] += wi * (double)kr[3 * i + 2];
        }
        block_sum<7, NW>(s7, scratch);
        const double inv = 1.0 / (s7[0] + a.kabsch_eps);
        const double k0 = s7[1] * inv, k1 = s7[2] * inv, k2 = s7[3] * inv;
        const double l0 = s7[4] * inv, l1 = s7[5] * inv, l2 = s7[6] * inv;
        double h9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = tid; i < nc; i += NT) {
            double wi = kw ? (double)kw[i] : 1.0;
            if (wi < a.kabsch_w_thresh) wi = 0.0;
            wi *= inv;
            const double a0 = (double)ks[3 * i] - k0, a1 = (double)ks[3 * i + 1] - k1, a2 = (double)ks[3 * i + 2] - k2;
            const double b0 = wi * ((double)kr[3 * i] - l0), b1 = wi * ((double)kr[3 * i + 1] - l1), b2 = wi * ((double)kr[3 * i + 2] - l2);
            h9[0] += a0 * b0; h9[1] += a0 * b1; h9[2] += a0 * b2;
            h9[3] += a1 * b0; h9[4] += a1 * b1; h9[5] += a1 * b2;
            h9[6] += a2 * b0; h9[7] += a2 * b1; h9[8] += a2 * b2;
        }
        __syncthreads();
        block_sum<9, NW>(h9, scratch);
        if (tid == 0 && nc > 0) {
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
        __syncthreads();
    }

    if (tid == 0) {
        if (fused_init || a.init_T) {
            const double *T = fused_init ? Tk : a.init_T + 16 * p;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                state[3 * i] = T[4 * i]; state[3 * i + 1] = T[4 * i + 1]; state[3 * i + 2] = T[4 * i + 2];
                state[9 + i] = T[4 * i] * (double)ox + T[4 * i + 1] * (double)oy + T[4 * i + 2] * (double)oz + T[4 * i + 3] -
                               (double)(i == 0 ? ox : (i == 1 ? oy : oz));
            }
        } else {
            state[0] = 1; state[1] = 0; state[2] = 0; state[3] = 0; state[4] = 1; state[5] = 0;
            state[6] = 0; state[7] = 0; state[8] = 1; state[9] = 0; state[10] = 0; state[11] = 0;
        }
        state[12] = 0.0; state[13] = 0.0; state[14] = 0.0; state[15] = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) state[16 + i] = (i % 4 == 0) ? 1.0 : 0.0;
        state[25] = cs[0]; state[26] = cs[1]; state[27] = cs[2]; state[28] = srad; state[29] = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) state[30 + i] = state[3 * i] * cs[0] + state[3 * i + 1] * cs[1] + state[3 * i + 2] * cs[2] + state[9 + i];
#pragma unroll
        for (int i = 0; i < LAZY_NM; ++i) M[i] = 0.0;
        state[33] = 1.0;  // (pass 0 searches every point anyway: its sum of squares is measured)
    }
    __syncthreads();

    if (active && g.wmax > 1) {
        PrepassArgs<F, NT> q;
        auto lds_off = [&](const void *ptr) { return (int)((const unsigned char *)ptr - smem_raw); };
        q.g = g; q.tl = lds_off(tl); q.E = lds_off(E); q.rl = lds_off(rl); q.sl = src_in_lds ? lds_off(sl) : -1;
        q.state = lds_off(state); q.prev = lds_off(prev); q.mabs = lds_off(mabs); q.ps = lds_off(ps);
        q.sg = sg; q.ox = ox; q.oy = oy; q.oz = oz; q.ns = ns; q.nt = nt; q.rs = rs;
        icp_prepass<F, NT>(q);
    }

    const int n_pass = active ? a.max_iter + 1 : 0;
    const int solver = (int)(blockIdx.x % NW);

    // a point's deadline from where it stands: e = distance to its correspondent (or nothing within the search radius), room =
    // the distance every OTHER target is known to keep from here, hit = the pair counts.  The certificate holds while the point has
    // moved D with e + D < room - D, the status while e + D < r (hit) or e - D >= r (a correspondent beyond the radius), or
    // room - D > r (no target within the search radius).  Rounded down, with the slack of icp_kernel's certificate tests.
    auto budget_of = [&](bool has, bool hit, F e, F room) -> F {
        F b;
        if (!has) b = room - rF * (F)1.000001;
        else {
            const F half = (room - e * (F)1.000002) * (F)0.5;
            const F stat = hit ? rF * (F)0.999999 - e * (F)1.000001 : e * (F)0.999999 - rF * (F)1.000001;
            b = half < stat ? half : stat;
        }
        b = b * (F)0.999 - (F)1e-9;
        return b > (F)0 ? b : (F)0;
    };
    for (int pass = 0; pass < n_pass; ++pass) {
        const F R0 = (F)uniform_f64(state[0]), R1 = (F)uniform_f64(state[1]), R2 = (F)uniform_f64(state[2]),
                R3 = (F)uniform_f64(state[3]), R4 = (F)uniform_f64(state[4]), R5 = (F)uniform_f64(state[5]),
                R6 = (F)uniform_f64(state[6]), R7 = (F)uniform_f64(state[7]), R8 = (F)uniform_f64(state[8]);
        const F t0f = (F)uniform_f64(state[9]), t1f = (F)uniform_f64(state[10]), t2f = (F)uniform_f64(state[11]);
        const F Bnow = (F)(uniform_f64(state[29]) * (1.0 + 1e-6));  // the motion bound so far, rounded up
        const float Bup = __double2float_ru(Bnow), Bdn = __double2float_rd(uniform_f64(state[29]));
        const bool direct = uniform_f64(state[33]) != 0.0;  // this pass measures sum d^2 point by point (header comment)
        const double c0 = cs[0], c1 = cs[1], c2 = cs[2];
        double acc[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i] = 0.0;
        bool touched = direct;  // this lane changed the moments in this pass (a direct pass reduces its sum of squares in any case)
        // sgn * the pair (origin-relative source point and target, taken about the source centroid) into the pass's deltas
        auto pair_delta = [&](double sgn, F x, F y, F z, F qx, F qy, F qz) {
            touched = true;
            x -= c0; y -= c1; z -= c2; qx -= c0; qy -= c1; qz -= c2;
            acc[0] += sgn;
            acc[1] += sgn * x; acc[2] += sgn * y; acc[3] += sgn * z;
            acc[4] += sgn * qx; acc[5] += sgn * qy; acc[6] += sgn * qz;
            const double sx = sgn * x, sy = sgn * y, sz = sgn * z;
            acc[7] += qx * sx; acc[8] += qx * sy; acc[9] += qx * sz;
            acc[10] += qy * sx; acc[11] += qy * sy; acc[12] += qy * sz;
            acc[13] += qz * sx; acc[14] += qz * sy; acc[15] += qz * sz;
            acc[16] += sgn * (x * x + y * y + z * z);
            acc[17] += sgn * (qx * qx + qy * qy + qz * qz);
        };
        auto source_of = [&](int i, F &x, F &y, F &z) {
            if (src_in_lds) { x = sl[3 * i]; y = sl[3 * i + 1]; z = sl[3 * i + 2]; }
            else { x = (F)sg[3 * i] - (F)ox; y = (F)sg[3 * i + 1] - (F)oy; z = (F)sg[3 * i + 2] - (F)oz; }
        };

        // ---- phase 1: the points whose deadline has passed are re-measured; certified again (a new deadline) or queued
        unsigned short *myq = queue + wave * seg;
        int nq = 0;
        for (int base = 0; base < ns; base += NT) {
            const int i = base + tid;
            const bool valid = i < ns;
            const int ii = valid ? i : ns - 1;
            const float dv = dl[ii];
            const bool counted = (__float_as_uint(dv) >> 31) != 0u;
            const bool due = valid && (direct || !(Bup < fabsf(dv)));
            bool need = false;
            if (due) {
                F x, y, z;
                source_of(ii, x, y, z);
                const F px = R0 * x + R1 * y + R2 * z + t0f;
                const F py = R3 * x + R4 * y + R5 * z + t1f;
                const F pz = R6 * x + R7 * y + R8 * z + t2f;
                const int pv = (int)prev[ii];  // 0xffff: never searched, 0xfffe: nothing within the search radius
                const F mx = px - (F)ps[3 * ii], my = py - (F)ps[3 * ii + 1], mz = pz - (F)ps[3 * ii + 2];
                F moved = grid_sqrt<F>(grid_d2(mx, my, mz)) * (F)1.000001;
                moved += (F)2e-7 * (fabs(px) + fabs(py) + fabs(pz));
                const F room = (F)mabs[ii] - moved;
                const GridPt<F> q = tl[pv < 0xfffe ? pv : 0];
                const F d = grid_d2(grid_query(px, g.ox) - grid_coord(q.x, px), grid_query(py, g.oy) - grid_coord(q.y, py),
                                    grid_query(pz, g.oz) - grid_coord(q.z, pz));
                const F e = grid_sqrt<F>(d);
                const bool cert = pv < 0xfffe ? e * (F)1.000001 < room : (pv == 0xfffe && room > rF * (F)1.000001);
                if (cert) {
                    const bool hit = pv < 0xfffe && d < r2;
                    if (direct && hit) acc[18] += d;
                    if (hit != counted) {
                        F qx, qy, qz;
                        grid_rel(g, q, qx, qy, qz);
                        pair_delta(hit ? 1.0 : -1.0, x, y, z, qx, qy, qz);
                    }
                    // from here: the certificate's base moves to where the point stands now
                    const F room_dn = room * (F)0.999999;
                    mabs[ii] = __double2float_rd(room_dn);
                    ps[3 * ii] = (float)px; ps[3 * ii + 1] = (float)py; ps[3 * ii + 2] = (float)pz;
                    // (ps is rounded to float32: the next re-measurement's `moved` carries the 2e-7 |p| allowance for exactly that)
                    const F bud = budget_of(pv < 0xfffe, hit, e, room_dn - (F)4e-7 * (fabs(px) + fabs(py) + fabs(pz)));
                    const float nd = __double2float_rd((F)Bdn + bud);
                    dl[ii] = hit ? -nd : nd;
                    if (hit && nd == 0.f) dl[ii] = __uint_as_float(0x80000000u);
                } else need = true;
            }
            const unsigned long long m = __ballot(need);
            if (need)
                myq[nq + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u))] = (unsigned short)i;
            nq += __builtin_popcountll(m);
        }
        if (lane == 0) qcnt[wave] = nq;
        __syncthreads();
        int n_search = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) n_search += qcnt[w];

        // ---- phase 2: search the queued points, 64 per wave (icp_kernel's search; the moments get the difference)
        for (int base = wave * 64; base < n_search; base += NT) {
            const int k = base + lane;
            const bool valid = k < n_search;
            int i = valid ? k : n_search - 1;
            {
                int w = 0, loc = i;
#pragma unroll
                for (int u = 0; u < NW - 1; ++u) {
                    const int c = qcnt[u];
                    const bool beyond = (w == u) && loc >= c;
                    loc = beyond ? loc - c : loc;
                    w = beyond ? w + 1 : w;
                }
                i = (int)queue[w * seg + loc];
            }
            F x, y, z;
            source_of(i, x, y, z);
            const F px = R0 * x + R1 * y + R2 * z + t0f;
            const F py = R3 * x + R4 * y + R5 * z + t1f;
            const F pz = R6 * x + R7 * y + R8 * z + t2f;
            Best<F> best;
            F b0 = rs2;
            const int pv = (int)prev[i];
            const bool counted = (__float_as_uint(dl[i]) >> 31) != 0u;
            if (pv < 0xfffe) {  // last pass's correspondence, re-measured, bounds the search from the start
                const GridPt<F> q = tl[pv];
                const F bb = grid_sqrt<F>(grid_d2(grid_query(px, g.ox) - grid_coord(q.x, px), grid_query(py, g.oy) - grid_coord(q.y, py),
                                                  grid_query(pz, g.oz) - grid_coord(q.z, pz))) * (F)1.000001 + mu;
                b0 = bb * bb < rs2 ? bb * bb : rs2;
            }
            best.init(b0);
            grid_nn<F, NT>(g, tl, nt, E, rl, valid, px, py, pz, best);
            const bool hit = valid && best.found() && best.d2() < r2;  // SearchHybrid: d2 < r^2
            if (direct && hit) acc[18] += best.d2();
            if (valid) {
                const int nslot = best.found() ? best.slot() : 0xfffe;
                if (counted && !(hit && nslot == pv)) {  // the old pair leaves the moments
                    F qx, qy, qz;
                    const GridPt<F> q = tl[pv];
                    grid_rel(g, q, qx, qy, qz);
                    pair_delta(-1.0, x, y, z, qx, qy, qz);
                }
                if (hit && !(counted && nslot == pv)) {  // the new one enters
                    F qx, qy, qz;
                    const GridPt<F> q = tl[nslot];
                    grid_rel(g, q, qx, qy, qz);
                    pair_delta(1.0, x, y, z, qx, qy, qz);
                }
                const F m2 = best.second < b0 ? best.second : b0;
                prev[i] = (unsigned short)nslot;
                const F room = grid_sqrt<F>(m2) * (F)0.999999;
                mabs[i] = __double2float_rd(room);
                ps[3 * i] = (float)px; ps[3 * i + 1] = (float)py; ps[3 * i + 2] = (float)pz;
                const F e = best.found() ? grid_sqrt<F>(best.d2()) : (F)0;
                const F bud = budget_of(best.found(), hit, e, room * (F)0.999999 - (F)4e-7 * (fabs(px) + fabs(py) + fabs(pz)));
                const float nd = __double2float_rd((F)Bdn + bud);
                dl[i] = hit ? (nd == 0.f ? __uint_as_float(0x80000000u) : -nd) : nd;
            }
        }

        // the pass's deltas: DPP reduction inside the wave, the NW partials through LDS (skipped by a workgroup nobody touched)
        const bool any_touched = __syncthreads_or(touched ? 1 : 0) != 0;
        if (any_touched) {
            double xs[NV / 4], ys[NV % 4 > 0 ? NV % 4 : 1];
            row_sums_transposed<NV, double>(acc, xs, ys);
            if ((lane & 12) == 0) {
                double *row = scratch + (wave * 4 + (lane >> 4)) * NV;
#pragma unroll
                for (int m = 0; m < NV / 4; ++m) row[4 * m + (lane & 3)] = xs[m];
                if ((lane & 3) == 0) {
#pragma unroll
                    for (int j = 0; j < NV % 4; ++j) row[4 * (NV / 4) + j] = ys[j];
                }
            }
            __syncthreads();
        }
        if (wave == solver) {
            __builtin_amdgcn_s_setprio(3);
            double Mv[LAZY_NM], dsum = 0.0;
            {
                const int vi = lane < NV ? lane : NV - 1;
                double t = vi < LAZY_NM ? M[vi] : 0.0;
                if (any_touched) {
#pragma unroll
                    for (int w = 0; w < SUM_ROWS; ++w) t += scratch[w * NV + vi];
                    if (lane < LAZY_NM) M[vi] = t;
                }
                const long long tb = __double_as_longlong(t);
                const int tlo = (int)(tb & 0xffffffffLL), thi = (int)(tb >> 32);
#pragma unroll
                for (int i = 0; i < LAZY_NM; ++i) {
                    const int lo = __builtin_amdgcn_readlane(tlo, i), hi = __builtin_amdgcn_readlane(thi, i);
                    Mv[i] = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
                }
                {
                    const int lo = __builtin_amdgcn_readlane(tlo, 18), hi = __builtin_amdgcn_readlane(thi, 18);
                    dsum = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
                }
            }
            // the pass's Umeyama quantities under the current transform, from the moments (header comment)
            const double Rc0[9] = {(double)R0, (double)R1, (double)R2, (double)R3, (double)R4, (double)R5, (double)R6, (double)R7, (double)R8};
            const double tc0[3] = {(double)t0f, (double)t1f, (double)t2f};
            const double tp[3] = {state[30], state[31], state[32]};  // t': the image of the source centroid under the current transform
            const double m = Mv[0];
            double Rsx[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) Rsx[i] = Rc0[3 * i] * Mv[1] + Rc0[3 * i + 1] * Mv[2] + Rc0[3 * i + 2] * Mv[3];
            double qRx[9], rh = 0.0;  // sum q_a (R x)_b, and <R, sum q x^T>
#pragma unroll
            for (int qa = 0; qa < 3; ++qa)
#pragma unroll
                for (int pb = 0; pb < 3; ++pb) {
                    qRx[3 * qa + pb] = Mv[7 + 3 * qa] * Rc0[3 * pb] + Mv[7 + 3 * qa + 1] * Rc0[3 * pb + 1] + Mv[7 + 3 * qa + 2] * Rc0[3 * pb + 2];
                    rh += Rc0[3 * qa + pb] * Mv[7 + 3 * qa + pb];
                }
            double sd;
            if (direct) sd = dsum;
            else {
                const double s0_ = tp[0] - c0, s1_ = tp[1] - c1, s2_ = tp[2] - c2;
                const double ss = s0_ * s0_ + s1_ * s1_ + s2_ * s2_;
                const double sRsx = s0_ * Rsx[0] + s1_ * Rsx[1] + s2_ * Rsx[2];
                const double ssq = s0_ * Mv[4] + s1_ * Mv[5] + s2_ * Mv[6];
                sd = (Mv[16] - 2.0 * rh + Mv[17]) + 2.0 * (sRsx - ssq) + m * ss;
                sd = sd > 0.0 ? sd : 0.0;
            }
            const double fitness = state[13], rmse = state[14];
            int iters = (int)state[15];
            const double fit_new = m > 0.0 ? m / (double)ns : 0.0;
            const double rmse_new = m > 0.0 ? sqrt(sd / m) : 0.0;
            bool done = false;
            if (pass > 0) {
                iters = pass;
                if (!a.fixed_iters && fabs(fitness - fit_new) < a.rel_fitness && fabs(rmse - rmse_new) < a.rel_rmse) done = true;
            }
            if (pass == a.max_iter) done = true;
            double Ru[9], tu[3];
            bool have = false;
            if (!done && m > 0.0) {
                have = true;
                // Eigen::umeyama without scaling (icp_kernel's solve): means and covariance from the centred moments
                const double im = fast_rcp(m);
                double sg9[9];
#pragma unroll
                for (int qa = 0; qa < 3; ++qa)
#pragma unroll
                    for (int pb = 0; pb < 3; ++pb) sg9[3 * qa + pb] = qRx[3 * qa + pb] * im - (Mv[4 + qa] * im) * (Rsx[pb] * im);
                if ((a.debug & 128) || !rot_newton(sg9, Ru)) {
                    double U[9], V[9], V0[9];
#pragma unroll
                    for (int i = 0; i < 9; ++i) V0[i] = state[16 + i];
                    svd3_warm(sg9, V0, U, V);
                    if (lane == 0) {
#pragma unroll
                        for (int i = 0; i < 9; ++i) state[16 + i] = V[i];
                    }
                    const double sgn = (det3(U) * det3(V) < 0.0) ? -1.0 : 1.0;
                    mul_diag_bt(U, sgn, V, Ru);
                }
                const double mp0 = Rsx[0] * im + tp[0], mp1 = Rsx[1] * im + tp[1], mp2 = Rsx[2] * im + tp[2];
                const double mq0 = Mv[4] * im + c0, mq1 = Mv[5] * im + c1, mq2 = Mv[6] * im + c2;
                tu[0] = mq0 - (Ru[0] * mp0 + Ru[1] * mp1 + Ru[2] * mp2);
                tu[1] = mq1 - (Ru[3] * mp0 + Ru[4] * mp1 + Ru[5] * mp2);
                tu[2] = mq2 - (Ru[6] * mp0 + Ru[7] * mp1 + Ru[8] * mp2);
            }
            if (lane == 0) {
                state[12] = done ? 1.0 : 0.0;
                state[13] = fit_new; state[14] = rmse_new; state[15] = (double)iters;
                // the expansion of sum d^2 is good to ~10 eps (sum |x|^2 + sum |q|^2): measured point by point where that would show in the
                // rmse beyond 1e-11 (m rmse <= 2e-4 of those sums)
                state[33] = (m > 0.0 && m * rmse_new <= 2e-4 * (Mv[16] + Mv[17])) ? 1.0 : 0.0;
            }
            if (have) {  // T <- update * T, and the bound on how far any source point moves with it (icp_kernel's)
                const double cp0 = state[30], cp1 = state[31], cp2 = state[32];
                double fro = 0.0;
#pragma unroll
                for (int i = 0; i < 9; ++i) { const double e = Ru[i] - ((i % 4 == 0) ? 1.0 : 0.0); fro += e * e; }
                const double n0 = Ru[0] * cp0 + Ru[1] * cp1 + Ru[2] * cp2 + tu[0];
                const double n1 = Ru[3] * cp0 + Ru[4] * cp1 + Ru[5] * cp2 + tu[1];
                const double n2 = Ru[6] * cp0 + Ru[7] * cp1 + Ru[8] * cp2 + tu[2];
                const double m0 = n0 - cp0, m1 = n1 - cp1, m2 = n2 - cp2;
                const double cpn = fast_sqrt(cp0 * cp0 + cp1 * cp1 + cp2 * cp2);
                const double eps_pos = 1e-14;
                double motion = fast_sqrt(fro) * state[28] + fast_sqrt(m0 * m0 + m1 * m1 + m2 * m2) + eps_pos * (state[28] + cpn);
                motion *= 1.0 + 1e-9;
                double Rn[9];
                mul3(Ru, Rc0, Rn);
                const double tn0 = Ru[0] * tc0[0] + Ru[1] * tc0[1] + Ru[2] * tc0[2] + tu[0];
                const double tn1 = Ru[3] * tc0[0] + Ru[4] * tc0[1] + Ru[5] * tc0[2] + tu[1];
                const double tn2 = Ru[6] * tc0[0] + Ru[7] * tc0[1] + Ru[8] * tc0[2] + tu[2];
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) state[i] = Rn[i];
                    state[9] = tn0; state[10] = tn1; state[11] = tn2;
                    state[29] += motion;
                    state[30] = n0; state[31] = n1; state[32] = n2;
                }
            }
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        const bool finished = state[12] != 0.0;
        if (finished) break;  // uniform across the workgroup
    }

    if (tid == 0) {
        const double o0 = ox, o1 = oy, o2 = oz;
        double Rc[9], tc[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) Rc[i] = state[i];
        tc[0] = state[9]; tc[1] = state[10]; tc[2] = state[11];
        const double fitness = state[13], rmse = state[14];
        const int iters = skipped ? -1 : (int)state[15];
        double *T = a.T_out + 16 * p;
#pragma unroll
        for (int i = 0; i < 3; ++i) { T[4 * i] = Rc[3 * i]; T[4 * i + 1] = Rc[3 * i + 1]; T[4 * i + 2] = Rc[3 * i + 2]; }
        T[3] = tc[0] - (Rc[0] * o0 + Rc[1] * o1 + Rc[2] * o2) + o0;
        T[7] = tc[1] - (Rc[3] * o0 + Rc[4] * o1 + Rc[5] * o2) + o1;
        T[11] = tc[2] - (Rc[6] * o0 + Rc[7] * o1 + Rc[8] * o2) + o2;
        T[12] = 0.0; T[13] = 0.0; T[14] = 0.0; T[15] = 1.0;
        if (a.fitness_out) a.fitness_out[p] = fitness;
        if (a.rmse_out) a.rmse_out[p] = rmse;
        if (a.iters_out) a.iters_out[p] = iters;
    }
    if (a.corr_out) {  // every point's correspondence as the last pass that measured the sums left it (utils/o3d_tools.py:64)
        for (int i = tid; i < ns; i += NT) {
            const bool counted = active && (__float_as_uint(dl[i]) >> 31) != 0u;
            a.corr_out[s0 + i] = counted ? (int)(tl[prev[i]].tag >> 16) : -1;
        }
    }
    if (a.rows_out && !skipped) {
        const double o0 = ox, o1 = oy, o2 = oz;
        double r[9], tr[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) r[i] = state[i];
        tr[0] = state[9] - (r[0] * o0 + r[1] * o1 + r[2] * o2) + o0;
        tr[1] = state[10] - (r[3] * o0 + r[4] * o1 + r[5] * o2) + o1;
        tr[2] = state[11] - (r[6] * o0 + r[7] * o1 + r[8] * o2) + o2;
        const int64_t w0 = a.rows_off ? a.rows_off[p] : s0;
        const int nrow = a.rows_off ? (int)(a.rows_off[p + 1] - w0) : ns;
        const float *__restrict__ wg = a.rows_src ? a.rows_src + 3 * w0 : sg;
        float *__restrict__ out6 = a.rows_out + 6 * w0;
        for (int i = tid; i < nrow; i += NT) {
            const float xf = wg[3 * i], yf = wg[3 * i + 1], zf = wg[3 * i + 2];
            const double x = xf, y = yf, z = zf;
            float *o6 = out6 + 6 * i;
            o6[0] = xf; o6[1] = yf; o6[2] = zf;
            o6[3] = (float)(r[0] * x + r[1] * y + r[2] * z + tr[0]);
            o6[4] = (float)(r[3] * x + r[4] * y + r[5] * z + tr[1]);
            o6[5] = (float)(r[6] * x + r[7] * y + r[8] * z + tr[2]);
        }
    }
}

}  // namespace f4l
