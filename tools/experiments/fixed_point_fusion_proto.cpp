// Prototype (host, single thread, test tooling): the reference's SEQUENTIAL greedy fusion
// (codelibrary/geometry/point_cloud/supervoxel_segmentation.h:117-176) computed as the fixed point of synchronous parallel
// iterations -- every centre evaluated against the previous iteration's estimate of what LOWER-indexed centres did -- and
// compared, label for label, with the plain sequential replay.  Reports the iterations each lambda round needs (the depth of
// the index-ordered dependency chains), i.e. whether a device version is worth building.
//   g++ -O2 -std=c++17 fixed_point_fusion_proto.cpp -o /tmp/fpf && /tmp/fpf case.bin
// case.bin: int32 n, int32 k, double resolution, float xyz[n][3], double nrm[n][3], int32 knn[n][k]
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace std;
static int n, k; static double res;
static vector<float> xyz; static vector<double> nrm; static vector<int32_t> knn;
static inline double metric(int a, int b) {
    const float *pa = &xyz[3 * (size_t)a], *pb = &xyz[3 * (size_t)b];
    const double *na = &nrm[3 * (size_t)a], *nb = &nrm[3 * (size_t)b];
    const double dot = na[0] * nb[0] + na[1] * nb[1] + na[2] * nb[2];
    const double t1 = (double)pa[0] - pb[0], t2 = (double)pa[1] - pb[1], t3 = (double)pa[2] - pb[2];
    return 1.0 - fabs(dot) + sqrt(t1 * t1 + t2 * t2 + t3 * t3) / res * 0.4;
}
static int occupied_cells() {
    double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (int i = 0; i < n; ++i) for (int d = 0; d < 3; ++d) { double v = xyz[3 * (size_t)i + d]; mn[d] = min(mn[d], v); mx[d] = max(mx[d], v); }
    int size[3]; for (int d = 0; d < 3; ++d) size[d] = (int)((mx[d] - mn[d]) / res + 1);
    vector<uint64_t> keys(n);
    for (int i = 0; i < n; ++i) { int c[3]; for (int d = 0; d < 3; ++d) { c[d] = (int)(((double)xyz[3 * (size_t)i + d] - mn[d]) / res); c[d] = min(max(c[d], 0), size[d] - 1); }
        keys[i] = ((uint64_t)c[0] * size[1] + c[1]) * size[2] + c[2]; }
    sort(keys.begin(), keys.end()); return (int)(unique(keys.begin(), keys.end()) - keys.begin());
}
struct Lists { vector<int64_t> off; vector<int32_t> len; vector<int32_t> pool; };
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); if (!f) return 1;
    if (fread(&n, 4, 1, f) != 1 || fread(&k, 4, 1, f) != 1 || fread(&res, 8, 1, f) != 1) return 1;
    xyz.resize(3 * (size_t)n); nrm.resize(3 * (size_t)n); knn.resize((size_t)n * k);
    if (fread(xyz.data(), 4, xyz.size(), f) != xyz.size() || fread(nrm.data(), 8, nrm.size(), f) != nrm.size() || fread(knn.data(), 4, knn.size(), f) != knn.size()) return 1;
    fclose(f);
    const int K = occupied_cells();
    vector<double> dis(n);
    for (int i = 0; i < n; ++i) { double b = DBL_MAX; for (int j = 0; j < k; ++j) { int q = knn[(size_t)i * k + j]; if (q != i) b = min(b, metric(i, q)); } dis[i] = b; }
    double lambda0; { vector<double> t(dis); nth_element(t.begin(), t.begin() + n / 2, t.end()); lambda0 = max(DBL_EPSILON, t[n / 2]); }
    // ---------------- sequential replay (the reference's order) ----------------
    vector<int32_t> par_s(n), size_s(n, 1), reps_s(n);
    {
        vector<vector<int32_t>> adj(n);
        for (int i = 0; i < n; ++i) { par_s[i] = i; reps_s[i] = i; adj[i].assign(&knn[(size_t)i * k], &knn[(size_t)i * k] + k); }
        auto find = [&](int x) { while (par_s[x] != x) { par_s[x] = par_s[par_s[x]]; x = par_s[x]; } return x; };
        vector<uint8_t> vis(n, 0); vector<int32_t> q(n); int live = n, nreps = n; double lambda = lambda0;
        for (;; lambda *= 2.0) {
            if (nreps <= 1) break;
            for (int s = 0; s < nreps; ++s) { int i = reps_s[s]; if (adj[i].empty()) continue;
                vis[i] = 1; int front = 0, back = 1; q[front++] = i;
                for (int a : adj[i]) { int j = find(a); if (!vis[j]) { vis[j] = 1; q[back++] = j; } }
                vector<int32_t> kept;
                while (front < back) { int j = q[front++];
                    if (lambda - size_s[j] * metric(i, j) > 0.0) { par_s[j] = i; size_s[i] += size_s[j];
                        for (int a : adj[j]) { int t = find(a); if (!vis[t]) { vis[t] = 1; q[back++] = t; } }
                        adj[j].clear(); if (--live == K) break; } else kept.push_back(j); }
                adj[i].swap(kept); for (int a = 0; a < back; ++a) vis[q[a]] = 0; if (live == K) break; }
            int m = 0; for (int s = 0; s < nreps; ++s) { int i = reps_s[s]; if (find(i) == i) reps_s[m++] = i; } nreps = m; live = m;
            if (nreps == K) break;
        }
        for (int i = 0; i < n; ++i) par_s[i] = find(i);
        printf("sequential: K target %d, reps %d\n", K, nreps);
    }
    // ---------------- the same by synchronous fixed-point iterations ----------------
    vector<int32_t> root(n), sz0(n, 1);             // committed state at the start of a round (root: flattened)
    vector<vector<int32_t>> adj0(n);                 // committed lists (ordered)
    for (int i = 0; i < n; ++i) { root[i] = i; adj0[i].assign(&knn[(size_t)i * k], &knn[(size_t)i * k] + k); }
    vector<int32_t> reps(n); for (int i = 0; i < n; ++i) reps[i] = i;
    int nreps = n, live = n; double lambda = lambda0; int round = 0; long total_iters = 0;
    const int32_t NONE = 0x7fffffff;
    vector<int32_t> abs_old(n, NONE), abs_new(n, NONE), ns_old(n), ns_new(n), cnt_old(n, 0), cnt_new(n, 0);
    vector<vector<int32_t>> kept_old(n), kept_new(n);
    vector<int64_t> before(n + 1);  // absorptions by lower centres (previous iteration's estimate), by position in reps
    vector<uint8_t> vis(n, 0); vector<int32_t> q(n);
    for (;; lambda *= 2.0, ++round) {
        if (nreps <= 1) break;
        for (int s = 0; s < nreps; ++s) { int i = reps[s]; abs_old[i] = NONE; ns_old[i] = sz0[i]; cnt_old[i] = 0; kept_old[i] = adj0[i]; }
        const int64_t budget_total = (int64_t)live - K;
        int iters = 0;
        for (;;) {
            ++iters;
            // absorptions by lower centres, from the previous iteration's counts
            before[0] = 0; for (int s = 0; s < nreps; ++s) before[s + 1] = before[s] + cnt_old[reps[s]];
            for (int s = 0; s < nreps; ++s) { int i = reps[s]; abs_new[i] = NONE; }
            bool changed = false;
            for (int s = 0; s < nreps; ++s) {  // (every centre independently: this loop is the parallel pass)
                const int i = reps[s];
                // Find as centre i sees it: the round-start root, then every absorption by a centre that ran BEFORE i -- a claim by c counts
                // only if c was alive at its own turn (not itself absorbed by a lower centre: such a c never ran); honoured claims
                // lead to ever higher centres, so the walk ends
                auto findv = [&](int a) { int r = root[a]; for (;;) { int c = abs_old[r];
                    if (c != NONE && c < i && !(abs_old[c] != NONE && abs_old[c] < c)) r = c; else break; } return r; };
                int nsz = sz0[i]; int cnt = 0; vector<int32_t> kept;
                const bool dead = abs_old[i] != NONE && abs_old[i] < i;
                int64_t budget = budget_total - before[s];
                if (!dead && !adj0[i].empty() && budget > 0) {
                    vis[i] = 1; int front = 0, back = 1; q[front++] = i;
                    for (int a : adj0[i]) { int j = findv(a); if (!vis[j]) { vis[j] = 1; q[back++] = j; } }
                    while (front < back) { int j = q[front++];
                        const int sj = j < i ? ns_old[j] : sz0[j];
                        if (lambda - sj * metric(i, j) > 0.0) {
                            abs_new[j] = min(abs_new[j], (int32_t)i); nsz += sj; ++cnt;
                            const vector<int32_t> &lj = j < i ? kept_old[j] : adj0[j];
                            for (int a : lj) { int t = findv(a); if (!vis[t]) { vis[t] = 1; q[back++] = t; } }
                            if (--budget == 0) break;
                        } else kept.push_back(j); }
                    for (int a = 0; a < back; ++a) vis[q[a]] = 0;
                } else if (!dead) kept = adj0[i];  // (did not run: its list stays)
                if (dead) kept.clear();
                if (nsz != ns_old[i] || cnt != cnt_old[i] || kept != kept_old[i]) changed = true;
                ns_new[i] = nsz; cnt_new[i] = cnt; kept_new[i].swap(kept);
            }
            for (int s = 0; s < nreps; ++s) { int i = reps[s]; if (abs_new[i] != abs_old[i]) changed = true; }
            for (int s = 0; s < nreps; ++s) { int i = reps[s]; abs_old[i] = abs_new[i]; ns_old[i] = ns_new[i]; cnt_old[i] = cnt_new[i]; kept_old[i].swap(kept_new[i]); }
            if (!changed) break;
            if (iters > 300) { printf("no convergence\n"); return 2; }
        }
        total_iters += iters;
        // commit
        int absorbed = 0;
        for (int s = 0; s < nreps; ++s) { int i = reps[s]; if (abs_old[i] != NONE) { ++absorbed; } }
        for (int s = 0; s < nreps; ++s) { int i = reps[s]; if (abs_old[i] == NONE) { sz0[i] = ns_old[i]; adj0[i] = kept_old[i]; } else adj0[i].clear(); }
        // roots: a node absorbed by c follows c (c may itself be absorbed by a later centre)
        vector<int32_t> newroot(n);
        auto final_root = [&](int r) { for (int hop = 0; abs_old[r] != NONE; ++hop) { if (hop > n) { printf("cycle\n"); exit(4); } r = abs_old[r]; } return r; };
        for (int i = 0; i < n; ++i) newroot[i] = final_root(root[i]);
        root.swap(newroot);
        int m = 0; for (int s = 0; s < nreps; ++s) { int i = reps[s]; if (abs_old[i] == NONE) reps[m++] = i; }
        printf("round %2d lambda %.4g: centres %8d absorbed %8d iterations %4d\n", round, lambda, nreps, absorbed, iters);
        nreps = m; live = m;
        if (nreps == K) break;
    }
    if (const char *d = getenv("FPF_DUMP")) { FILE *o = fopen(d, "wb"); fwrite(root.data(), 4, n, o); fclose(o); }  // (the roots: input of fixed_point_exchange_proto)
    long bad = 0; for (int i = 0; i < n; ++i) bad += root[i] != par_s[i];
    printf("fixed point: reps %d, total iterations %ld, labels differing from the sequential replay: %ld\n", nreps, total_iters, bad);
    return bad ? 3 : 0;
}
