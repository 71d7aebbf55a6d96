// Prototype (host, test tooling): the reference's boundary exchange (supervoxel_segmentation.h:186-237, a FIFO work list) computed
// generation by generation -- the entries the queue holds when a generation starts -- each generation as the fixed point of
// synchronous parallel iterations (an entry sees the outcome of EARLIER entries of its generation, as last estimated), the next
// generation's queue from the minimum (entry position, neighbour slot) that pushes a point.  Compared label for label with the
// sequential FIFO.  Reports generations and iterations.
//   g++ -O2 -std=c++17 fixed_point_exchange_proto.cpp -o /tmp/fpx && /tmp/fpx case.bin labels.bin   (labels.bin: int32 n roots after the fusion)
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
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); if (!f) return 1;
    if (fread(&n, 4, 1, f) != 1 || fread(&k, 4, 1, f) != 1 || fread(&res, 8, 1, f) != 1) return 1;
    xyz.resize(3 * (size_t)n); nrm.resize(3 * (size_t)n); knn.resize((size_t)n * k);
    if (fread(xyz.data(), 4, xyz.size(), f) != xyz.size() || fread(nrm.data(), 8, nrm.size(), f) != nrm.size() || fread(knn.data(), 4, knn.size(), f) != knn.size()) return 1;
    fclose(f);
    vector<int32_t> lab_in(n);
    f = fopen(argv[2], "rb"); if (!f || fread(lab_in.data(), 4, n, f) != (size_t)n) return 1; fclose(f);
    // ---- sequential FIFO
    vector<int32_t> ls(lab_in); vector<double> ds(n);
    {
        for (int i = 0; i < n; ++i) ds[i] = metric(i, ls[i]);
        vector<int32_t> fifo(n); vector<uint8_t> inq(n, 0); int64_t head = 0, tail = 0, count = 0;
        auto push = [&](int v) { fifo[tail] = v; tail = tail + 1 == n ? 0 : tail + 1; ++count; inq[v] = 1; };
        for (int i = 0; i < n; ++i) for (int j = 0; j < k; ++j) { int q = knn[(size_t)i * k + j]; if (ls[i] != ls[q]) { if (!inq[i]) push(i); if (!inq[q]) push(q); } }
        long pops = 0;
        while (count > 0) { int i = fifo[head]; head = head + 1 == n ? 0 : head + 1; --count; inq[i] = 0; ++pops; bool ch = false;
            for (int j = 0; j < k; ++j) { int q = knn[(size_t)i * k + j]; int a = ls[i], b = ls[q]; if (a == b) continue; double d = metric(i, b); if (d < ds[i]) { ls[i] = b; ds[i] = d; ch = true; } }
            if (ch) for (int j = 0; j < k; ++j) { int q = knn[(size_t)i * k + j]; if (ls[i] != ls[q] && !inq[q]) push(q); } }
        printf("sequential: %ld pops\n", pops);
    }
    // ---- generations
    vector<int32_t> lab(lab_in), out_lab(n), new_lab(n); vector<double> dis(n), out_dis(n), new_dis(n); vector<uint8_t> out_ch(n, 0), new_ch(n, 0);
    for (int i = 0; i < n; ++i) dis[i] = metric(i, lab[i]);
    const uint64_t INF = ~0ULL;
    vector<uint64_t> key(n, INF);
    // first generation: the scan of :194-207 -- a point enters at the first event that touches it
    for (int i = 0; i < n; ++i) for (int j = 0; j < k; ++j) { int q = knn[(size_t)i * k + j]; if (lab[i] != lab[q]) {
        key[i] = min(key[i], ((uint64_t)i * 64 + j) * 2); key[q] = min(key[q], ((uint64_t)i * 64 + j) * 2 + 1); } }
    vector<int32_t> Q; vector<int32_t> pos(n, 0x7fffffff);
    auto build_queue = [&]() { vector<pair<uint64_t, int32_t>> v; for (int x = 0; x < n; ++x) if (key[x] != INF) v.push_back({key[x], x}); sort(v.begin(), v.end());
        Q.clear(); for (auto &p : v) Q.push_back(p.second); for (int x = 0; x < n; ++x) { pos[x] = 0x7fffffff; key[x] = INF; } for (size_t t = 0; t < Q.size(); ++t) pos[Q[t]] = (int)t; };
    build_queue();
    int gens = 0; long iters_total = 0, pops = 0; int max_it = 0;
    while (!Q.empty()) {
        ++gens; pops += (long)Q.size();
        const int m = (int)Q.size();
        for (int t = 0; t < m; ++t) { int i = Q[t]; out_lab[i] = lab[i]; out_dis[i] = dis[i]; out_ch[i] = 0; }
        int it = 0;
        for (;;) { ++it; bool changed = false;
            for (int t = 0; t < m; ++t) {  // (parallel pass)
                int i = Q[t]; int a = lab[i]; double d0 = dis[i]; bool ch = false;
                for (int j = 0; j < k; ++j) { int q = knn[(size_t)i * k + j]; int b = pos[q] < t ? out_lab[q] : lab[q]; if (q == i) b = a;
                    if (a == b) continue; double d = metric(i, b); if (d < d0) { a = b; d0 = d; ch = true; } }
                new_lab[i] = a; new_dis[i] = d0; new_ch[i] = ch; }
            for (int t = 0; t < m; ++t) { int i = Q[t]; if (new_lab[i] != out_lab[i] || new_ch[i] != out_ch[i]) changed = true; out_lab[i] = new_lab[i]; out_dis[i] = new_dis[i]; out_ch[i] = new_ch[i]; }
            if (!changed) break; if (it > 10000) { printf("no convergence\n"); return 2; } }
        iters_total += it; max_it = max(max_it, it);
        // pushes of the generation (:228-236): by changed entries, of neighbours whose label differs and that are not in the queue
        for (int t = 0; t < m; ++t) { int i = Q[t]; if (!out_ch[i]) continue;
            for (int j = 0; j < k; ++j) { int q = knn[(size_t)i * k + j]; int b = q == i ? out_lab[i] : (pos[q] < t ? out_lab[q] : lab[q]);
                if (out_lab[i] != b && !(pos[q] != 0x7fffffff && pos[q] > t)) key[q] = min(key[q], (uint64_t)t * 64 + j); } }
        for (int t = 0; t < m; ++t) { int i = Q[t]; lab[i] = out_lab[i]; dis[i] = out_dis[i]; }
        build_queue();
    }
    long bad = 0; for (int i = 0; i < n; ++i) bad += lab[i] != ls[i];
    printf("generations %d, pops %ld, iterations in all %ld (largest %d), labels differing from the sequential FIFO: %ld\n", gens, pops, iters_total, max_it, bad);
    return bad ? 3 : 0;
}
