// ablation of k_mask_annotate phases on uniform synthetic rows
#define HINGE_ABLATE 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <cmath>
#include <cstdlib>
#include "../hinge_amd/csrc/filter_kernels.h"
using namespace hinge;
template <typename F> float timeit(F f, int reps = 10) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < reps; i++) f();
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}
int main() {
    const int nr = 87400;
    const bool varlen = getenv("MA_VARLEN") != nullptr;
    std::mt19937 rng(1);
    std::lognormal_distribution<double> ln(std::log(8500.0) - 0.5 * 0.35 * 0.35, 0.35);
    std::vector<int64_t> h(nr + 1);
    std::vector<int> hl(nr);
    h[0] = 0;
    int maxrl = 0;
    for (int i = 0; i < nr; i++) {
        int rl = varlen ? std::min(40000, std::max(1500, (int)ln(rng))) : 8500;
        hl[i] = rl; maxrl = std::max(maxrl, rl);
        int per = varlen ? (int)(300.0 * (rl + 7000.0) / 15500.0) : 300;
        h[i + 1] = h[i] + per;
    }
    const long n = h[nr];
    std::vector<int2> ha(n);
    for (int r = 0; r < nr; r++) {
        const int rl = hl[r];
        for (long i = h[r]; i < h[r + 1]; i++) {
            int len = 1000 + rng() % 7000; if (len > rl) len = rl;
            int ab = (rng() & 1) ? (int)(rng() % 26) : (int)(rng() % (rl - len + 1)); int ae = std::min(rl, ab + len);
            if (rng() & 1) ae = rl - (int)(rng() % 26); if (ae - ab < 500) ab = 0; ha[i] = make_int2(ab, ae);
        }
    }
    printf("overlaps %ld, max rlen %d\n", n, maxrl);
    int2* a; int64_t* rp; int* rl; int2 *mask, *cmask, *anno; unsigned char *rf, *hf; unsigned *aoff, *cnt; int *acnt, *st, *mc; WorkItem* wl;
    (void)hipMalloc(&a, n * 8); (void)hipMalloc(&rp, (nr + 1) * 8); (void)hipMalloc(&rl, nr * 4); (void)hipMalloc(&mask, nr * 8); (void)hipMalloc(&cmask, nr * 8);
    (void)hipMalloc(&anno, 4 * nr * 8); (void)hipMalloc(&rf, nr); (void)hipMalloc(&hf, 4 * nr); (void)hipMalloc(&aoff, nr * 4); (void)hipMalloc(&acnt, nr * 4);
    (void)hipMalloc(&wl, (size_t)nr * sizeof(WorkItem)); (void)hipMalloc(&cnt, 16); (void)hipMalloc(&st, 4); (void)hipMalloc(&mc, 4);
    (void)hipMemcpy(rp, h.data(), (nr + 1) * 8, hipMemcpyHostToDevice); (void)hipMemcpy(rl, hl.data(), nr * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(a, ha.data(), n * 8, hipMemcpyHostToDevice);
    int minc = 50; (void)hipMemcpy(mc, &minc, 4, hipMemcpyHostToDevice);
    FilterDev P{}; P.reso = 40; P.cut_off = 300; P.theta = 300; P.cov_frac = 3; P.min_ra = 10; P.max_ra = 20; P.ra_gap = 300; P.nhr = 500;
    P.sup = 7; P.pil = 7; P.unb = 6; P.tol = 100; P.bin_len = 200; P.use_qv = 0; P.use_cov = 1; P.del_telo = 0;
    const int kcap = ((maxrl + 300) / 40 + 4 + 3) & ~3;
    const size_t lds = 4 * 2 * kcap * 4;
    (void)hipFuncSetAttribute((const void*)k_mask_annotate<40>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int slot0 = ((((std::min(maxrl, 16000)) / 20 + 1 + 3) & ~3) + 4) + 4 * 64;
    const int slot = slot0 + 20 + 36;   // + zero / total pads for cut_off 300
    const size_t lds20 = 4 * (size_t)slot * 4;
    const int len1 = (slot0 - 4 * 64 - 1) * 20 + 19, len2 = (2 * slot0 - 4 * 64 - 1) * 20 + 19;
    std::vector<int> l1, l2, l4;
    for (int i = 0; i < nr; i++) (hl[i] <= len1 ? l1 : hl[i] <= len2 ? l2 : l4).push_back(i);
    std::vector<int> lst(l1); lst.insert(lst.end(), l2.begin(), l2.end()); lst.insert(lst.end(), l4.begin(), l4.end());
    const int n1 = (int)l1.size(), n2 = (int)l2.size(), n4 = (int)l4.size();
    printf("classes: %d / %d / %d, slot %d ints, LDS %zu B per workgroup\n", n1, n2, n4, slot, lds20);
    int* fb; (void)hipMalloc(&fb, nr * 4);
    int* ids; (void)hipMalloc(&ids, nr * 4);
    (void)hipMemcpy(ids, lst.data(), nr * 4, hipMemcpyHostToDevice);
    // 16|16 copy (+ one wavefront of padding), bin counts from k_cov_stats
    std::vector<unsigned> h16(n + 256, 0u);
    for (long i = 0; i < n; i++) h16[i] = (unsigned)ha[i].x | ((unsigned)ha[i].y << 16);
    unsigned* a16; (void)hipMalloc(&a16, (n + 256) * 4); (void)hipMemcpy(a16, h16.data(), (n + 256) * 4, hipMemcpyHostToDevice);
    int *meanc, *nb0, *psc; unsigned long long* wt;
    (void)hipMalloc(&meanc, nr * 4); (void)hipMalloc(&nb0, nr * 4); (void)hipMalloc(&psc, 64 * 4); (void)hipMalloc(&wt, 2048 * 4 * 2 * 8);
    float tk1 = timeit([&] { hipLaunchKernelGGL((k_cov_stats<40, true>), dim3(2048), dim3(256), 0, 0, 0, nr - 1, rp, a, a16, rl, 40, meanc, nb0, wt, psc, 16, mc, 0, 0); });
    printf("k_cov_stats<40, packed>: %7.1f us\n", tk1 * 1e3);
    AnnoOut o{nullptr, nullptr, mask, cmask, rf, anno, hf, aoff, acnt, cnt, 4u * (unsigned)nr, wl, st};
    const int grid = (nr + 3) / 4;
    const int rpw = getenv("MA_RPW") ? atoi(getenv("MA_RPW")) : 3;
    const int grid20 = ((n1 + 3) / 4 + rpw - 1) / rpw + (n2 + 1) / 2 + n4;
    for (int mode = 1; mode <= 8; mode++) {
        P.ablate = mode;
        float t = timeit([&] { (void)hipMemsetAsync(cnt, 0, 16, 0);
            hipLaunchKernelGGL(k_mask_annotate<40>, dim3(grid), dim3(256), lds, 0, P, 0, nr - 1, rp, a, rl, mc, kcap, o, (const int*)nullptr, (const unsigned*)nullptr); });
        float t2 = timeit([&] { (void)hipMemsetAsync(cnt, 0, 16, 0);
            hipLaunchKernelGGL(k_mask_annotate_q20<false>, dim3(grid20), dim3(256), lds20, 0, P, ids, n1, n2, n4, rp, a, rl, nb0, mc, slot, o, fb, cnt + 2, rpw); });
        float t3 = timeit([&] { (void)hipMemsetAsync(cnt, 0, 16, 0);
            hipLaunchKernelGGL(k_mask_annotate_q20<true>, dim3(grid20), dim3(256), lds20, 0, P, ids, n1, n2, n4, rp, a16, rl, nb0, mc, slot, o, fb, cnt + 2, rpw); });
        printf("stop after phase %d: general %7.1f us   q20 %7.1f us   q20 packed %7.1f us   (1 histogram, 2 +mask, 3 +gate, 4 +candidates, 5 all; to end of phase 2: 6 no scan, 7 no mask pass, 8 neither)\n", mode, t * 1e3, t2 * 1e3, t3 * 1e3);
    }
    unsigned hc[4]; (void)hipMemcpy(hc, cnt, 16, hipMemcpyDeviceToHost);
    printf("counters: anno %u work %u fallback %u\n", hc[0], hc[1], hc[2]);
    return 0;
}
