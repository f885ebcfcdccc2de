// Native smoke/diagnostic for the C ABI: random non-negative CSR x CSR^T, top-n, compared with a plain
// CPU loop.  Build: hipcc -O2 tests/native/k4_smoke.cpp -Iinclude -Lstring_grouper_amd -lsg_hip -o /tmp/k4_smoke
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "sg_hip.h"

#include <dlfcn.h>
#include <math.h>

int main(int argc, char **argv) {
    const int64_t nL = argc > 1 ? atoll(argv[1]) : 2000, nR = argc > 2 ? atoll(argv[2]) : 3000, V = 400;
    const int top_n = 5;
    const float thr = 0.3f;
    srand(1);
    auto gen = [&](int64_t n, std::vector<int64_t> &ip, std::vector<int32_t> &ix, std::vector<float> &d) {
        ip.assign(1, 0);
        for (int64_t i = 0; i < n; ++i) {
            int k = 3 + rand() % 20;
            std::vector<int> cols;
            for (int q = 0; q < k; ++q) cols.push_back(rand() % (rand() % 4 == 0 ? 8 : (int)V));   // a few heavy columns
            std::sort(cols.begin(), cols.end());
            cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
            float nrm = 0;
            std::vector<float> vals;
            for (size_t q = 0; q < cols.size(); ++q) { float v = 0.1f + (rand() % 100) / 100.f; vals.push_back(v); nrm += v * v; }
            nrm = sqrtf(nrm);
            for (size_t q = 0; q < cols.size(); ++q) { ix.push_back(cols[q]); d.push_back(vals[q] / nrm); }
            ip.push_back((int64_t)ix.size());
        }
    };
    std::vector<int64_t> aip, bip; std::vector<int32_t> aix, bix; std::vector<float> ad, bd;
    gen(nL, aip, aix, ad); gen(nR, bip, bix, bd);
    sg_ctx *ctx = nullptr;
    if (sg_ctx_create(0, nullptr, &ctx)) { printf("ctx: %s\n", sg_last_error()); return 2; }
    std::vector<int32_t> oc(nL * top_n), cnt(nL); std::vector<float> ov(nL * top_n);
    int rc = sg_sp_matmul_topn_host(ctx, nL, nR, V, aip.data(), aix.data(), ad.data(), bip.data(), bix.data(), bd.data(),
                                    SG_F32, top_n, thr, 1, oc.data(), ov.data(), cnt.data());
    printf("rc=%d %s\n", rc, rc ? sg_last_error() : "");
    for (const char *sym : {"sg_debug_watch", "sg_debug_watch_pruned"})   // only in -DSG_WATCHDOG builds
        if (auto fn = (int (*)(int32_t *))dlsym(RTLD_DEFAULT, sym)) {
            int32_t w[4] = {0, 0, 0, 0};
            fn(w);
            printf("%s: code %d count %d\n", sym, w[0], w[1]);
        }
    sg_stats st; sg_ctx_stats(ctx, &st);
    printf("K4 %.3f ms, macs %lld, out %lld; pruned rows %lld postings %lld survivors %lld, exact rows %lld\n",
           st.ms[SG_K_SPGEMM], (long long)st.macs, (long long)st.out_nnz, (long long)st.prune_rows,
           (long long)st.prune_postings, (long long)st.prune_survivors, (long long)st.exact_rows);
    // CPU check
    long bad = 0;
    std::vector<float> acc(nR);
    // Bt
    std::vector<std::vector<std::pair<int,float>>> post(V);
    for (int64_t j = 0; j < nR; ++j) for (int64_t p = bip[j]; p < bip[j+1]; ++p) post[bix[p]].push_back({(int)j, bd[p]});
    for (int64_t i = 0; i < nL; ++i) {
        std::fill(acc.begin(), acc.end(), 0.f);
        for (int64_t p = aip[i]; p < aip[i+1]; ++p) for (auto &e : post[aix[p]]) { volatile float pr = ad[p] * e.second; acc[e.first] = acc[e.first] + pr; }
        std::vector<std::pair<float,int>> c;
        for (int64_t j = 0; j < nR; ++j) if (acc[j] > thr) c.push_back({-acc[j], (int)j});
        std::sort(c.begin(), c.end());
        if (c.size() > (size_t)top_n) c.resize(top_n);
        if ((int)c.size() != cnt[i]) { if (bad < 5) printf("row %lld count %d vs %zu\n", (long long)i, cnt[i], c.size()); ++bad; continue; }
        for (size_t q = 0; q < c.size(); ++q)
            if (oc[i*top_n+q] != c[q].second || ov[i*top_n+q] != -c[q].first) { if (bad < 5) printf("row %lld slot %zu: %d %.9g vs %d %.9g\n", (long long)i, q, oc[i*top_n+q], ov[i*top_n+q], c[q].second, -c[q].first); ++bad; }
    }
    printf("mismatches: %ld\n", bad);
    sg_ctx_destroy(ctx);
    return bad ? 1 : 0;
}
