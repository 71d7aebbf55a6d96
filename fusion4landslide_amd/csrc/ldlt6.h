// ldlt6.h -- x = A.ldlt().solve(b) for a symmetric 6 x 6 the way Eigen computes it, in registers.
//
// Open3D's point-to-plane step ends in utility::SolveLinearSystemPSD(JTJ, -JTr) with its checks off, i.e.
// `JTJ.ldlt().solve(-JTr)`, always reported as solved (utils/o3d_tools.py:38-39,46-50 select the estimator;
// [3P-knowledge]: Open3D 0.19 / Eigen 3.4 are not under /root/reference).  Eigen::LDLT is the bordered, left-looking
// factorisation P A P^T = L D L^T: the pivot of step k is the largest |diagonal entry| among the rows not yet eliminated
// AS STORED (the trailing block is never updated, so these are entries of A itself), first index on ties; a pivot that is
// exactly zero leaves its column undivided; solve() applies the pseudo-inverse of D with tolerance DBL_MIN.  The same
// statements, with loops over memory, are oracle/f4l_oracle.c::orc_ldlt6_solve_eigen; a CPU test holds the two together
// (tests/test_oracle_icp.py builds this header as host code).
//
// Every index below is a compile-time constant after unrolling (the data-dependent transpositions are predicated
// swaps of fixed register pairs), so nothing goes to scratch memory.
#pragma once

// Eigen's result depends on the last bits of these products and sums (which diagonal entry is the pivot, whether a pivot
// passes the DBL_MIN test of the pseudo-inverse): no fused multiply-adds in this header on the device either (ADVICE r4;
// hipcc contracts by default, the host build of the tests uses -ffp-contract=off).
#ifdef __clang__
#define F4L_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define F4L_NO_CONTRACT
#endif

#ifndef F4L_HD
#ifdef __HIPCC__
#define F4L_HD __host__ __device__ __forceinline__
#else
#define F4L_HD inline
#endif
#endif

namespace f4l {

F4L_HD void ldlt6_swap(double &a, double &b) { const double t = a; a = b; b = t; }
F4L_HD double ldlt6_abs(double v) { return v < 0.0 ? -v : v; }

// symmetric transposition k <-> c (k < c) of the lower triangle
template <int K, int C> F4L_HD void ldlt6_transpose(double (&A)[6][6]) {
#pragma unroll
    for (int j = 0; j < K; ++j) ldlt6_swap(A[K][j], A[C][j]);
#pragma unroll
    for (int i = C + 1; i < 6; ++i) ldlt6_swap(A[i][K], A[i][C]);
    ldlt6_swap(A[K][K], A[C][C]);
#pragma unroll
    for (int i = K + 1; i < C; ++i) ldlt6_swap(A[i][K], A[C][i]);
}

template <int K> F4L_HD void ldlt6_step(double (&A)[6][6], int (&tr)[6], bool &whole_diagonal_zero) {
    F4L_NO_CONTRACT
    if (whole_diagonal_zero) return;
    int big = K;
    double best = ldlt6_abs(A[K][K]);
#pragma unroll
    for (int i = K + 1; i < 6; ++i) {
        const double v = ldlt6_abs(A[i][i]);
        if (v > best) { best = v; big = i; }
    }
    tr[K] = big;
    if constexpr (K < 1) { if (big == 1) ldlt6_transpose<K, 1>(A); }
    if constexpr (K < 2) { if (big == 2) ldlt6_transpose<K, 2>(A); }
    if constexpr (K < 3) { if (big == 3) ldlt6_transpose<K, 3>(A); }
    if constexpr (K < 4) { if (big == 4) ldlt6_transpose<K, 4>(A); }
    if constexpr (K < 5) { if (big == 5) ldlt6_transpose<K, 5>(A); }
    if constexpr (K > 0) {
        double temp[6];
#pragma unroll
        for (int j = 0; j < K; ++j) temp[j] = A[j][j] * A[K][j];
        double dot = 0.0;
#pragma unroll
        for (int j = 0; j < K; ++j) dot += A[K][j] * temp[j];
        A[K][K] -= dot;
#pragma unroll
        for (int i = K + 1; i < 6; ++i) {
            double d = 0.0;
#pragma unroll
            for (int j = 0; j < K; ++j) d += A[i][j] * temp[j];
            A[i][K] -= d;
        }
    }
    const double akk = A[K][K];
    const bool valid = ldlt6_abs(akk) > 0.0;
    if (K == 0 && !valid) {  // the whole diagonal is zero: nothing is factorised, the transpositions are the identity
        whole_diagonal_zero = true;
        return;
    }
    if (valid) {
#pragma unroll
        for (int i = K + 1; i < 6; ++i) A[i][K] /= akk;
    }
}

template <int K> F4L_HD void ldlt6_permute(double (&y)[6], const int (&tr)[6]) {
    if constexpr (K < 1) { if (tr[K] == 1) ldlt6_swap(y[K], y[1]); }
    if constexpr (K < 2) { if (tr[K] == 2) ldlt6_swap(y[K], y[2]); }
    if constexpr (K < 3) { if (tr[K] == 3) ldlt6_swap(y[K], y[3]); }
    if constexpr (K < 4) { if (tr[K] == 4) ldlt6_swap(y[K], y[4]); }
    if constexpr (K < 5) { if (tr[K] == 5) ldlt6_swap(y[K], y[5]); }
}

// A: symmetric, read in its lower triangle and destroyed; x = A.ldlt().solve(b).
F4L_HD void ldlt6_solve_eigen(double (&A)[6][6], const double (&b)[6], double (&x)[6]) {
    F4L_NO_CONTRACT
    int tr[6] = {0, 1, 2, 3, 4, 5};
    bool zero_diag = false;
    ldlt6_step<0>(A, tr, zero_diag);
    ldlt6_step<1>(A, tr, zero_diag);
    ldlt6_step<2>(A, tr, zero_diag);
    ldlt6_step<3>(A, tr, zero_diag);
    ldlt6_step<4>(A, tr, zero_diag);
    ldlt6_step<5>(A, tr, zero_diag);
    if (zero_diag) {
#pragma unroll
        for (int i = 0; i < 6; ++i) tr[i] = i;
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] = b[i];
    ldlt6_permute<0>(y, tr); ldlt6_permute<1>(y, tr); ldlt6_permute<2>(y, tr);  // P b
    ldlt6_permute<3>(y, tr); ldlt6_permute<4>(y, tr);
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < i; ++j) y[i] -= A[i][j] * y[j];  // L^-1
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] = ldlt6_abs(A[i][i]) > 2.2250738585072014e-308 ? y[i] / A[i][i] : 0.0;  // D^+
#pragma unroll
    for (int i = 5; i >= 0; --i)
#pragma unroll
        for (int j = i + 1; j < 6; ++j) y[i] -= A[j][i] * y[j];  // L^-T
    ldlt6_permute<4>(y, tr); ldlt6_permute<3>(y, tr); ldlt6_permute<2>(y, tr);  // P^T
    ldlt6_permute<1>(y, tr); ldlt6_permute<0>(y, tr);
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = y[i];
}

// The 6 x 6 point-to-plane system accumulated about the patch origin o (J' = [(s - o) x n, n]) expressed about the
// CALLER's origin, the frame Open3D builds it in: J = [s x n, n] = C J' with C = [[I, [o]x], [0, I]], hence
// M = C M' C^T, b = C b'.  M' symmetric (full storage), in place.
F4L_HD void p2plane_system_to_caller_frame(double (&M)[6][6], double (&b)[6], double o0, double o1, double o2) {
    F4L_NO_CONTRACT
    const double K[3][3] = {{0.0, -o2, o1}, {o2, 0.0, -o0}, {-o1, o0, 0.0}};
    double KBt[3][3], KD[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double s = 0.0, t = 0.0;
#pragma unroll
            for (int l = 0; l < 3; ++l) { s += K[i][l] * M[j][3 + l]; t += K[i][l] * M[3 + l][3 + j]; }  // K B^T, K D
            KBt[i][j] = s; KD[i][j] = t;
        }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double kdk = 0.0;
#pragma unroll
            for (int l = 0; l < 3; ++l) kdk += KD[i][l] * K[j][l];  // K D K^T
            M[i][j] += KBt[i][j] + KBt[j][i] + kdk;
        }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) { M[i][3 + j] += KD[i][j]; M[3 + j][i] = M[i][3 + j]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) b[i] += K[i][0] * b[3] + K[i][1] * b[4] + K[i][2] * b[5];
}

}  // namespace f4l
