// ablation of k_mask_annotate phases on uniform synthetic rows
#define HINGE_ABLATE 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
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
    const int nr = 87400, per = 300; const long n = (long)nr * per;
    std::mt19937 rng(1);
    std::vector<int64_t> h(nr + 1); for (int i = 0; i <= nr; i++) h[i] = (int64_t)i * per;
    std::vector<int> hl(nr, 8500);
    std::vector<int2> ha(n);
    for (long i = 0; i < n; i++) { int len = 1000 + rng() % 7000; int ab = (rng() & 1) ? (int)(rng() % 26) : (int)(rng() % (8500 - len)); int ae = std::min(8500, ab + len); if (rng() & 1) ae = 8500 - (int)(rng() % 26); if (ae - ab < 500) ab = 0; ha[i] = make_int2(ab, ae); }
    int2* a; int64_t* rp; int* rl; int2 *mask, *cmask, *anno; unsigned char *rf, *hf; unsigned *aoff, *cnt; int *acnt, *wl, *st, *mc;
    (void)hipMalloc(&a, n * 8); (void)hipMalloc(&rp, (nr + 1) * 8); (void)hipMalloc(&rl, nr * 4); (void)hipMalloc(&mask, nr * 8); (void)hipMalloc(&cmask, nr * 8);
    (void)hipMalloc(&anno, 4 * nr * 8); (void)hipMalloc(&rf, nr); (void)hipMalloc(&hf, 4 * nr); (void)hipMalloc(&aoff, nr * 4); (void)hipMalloc(&acnt, nr * 4);
    (void)hipMalloc(&wl, nr * 4); (void)hipMalloc(&cnt, 16); (void)hipMalloc(&st, 4); (void)hipMalloc(&mc, 4);
    (void)hipMemcpy(rp, h.data(), (nr + 1) * 8, hipMemcpyHostToDevice); (void)hipMemcpy(rl, hl.data(), nr * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(a, ha.data(), n * 8, hipMemcpyHostToDevice);
    int minc = 50; (void)hipMemcpy(mc, &minc, 4, hipMemcpyHostToDevice);
    FilterDev P{}; P.reso = 40; P.cut_off = 300; P.theta = 300; P.cov_frac = 3; P.min_ra = 10; P.max_ra = 20; P.ra_gap = 300; P.nhr = 500;
    P.sup = 7; P.pil = 7; P.unb = 6; P.tol = 100; P.bin_len = 200; P.use_qv = 0; P.use_cov = 1; P.del_telo = 0;
    const int kcap = ((8500 + 300) / 40 + 4 + 3) & ~3;
    const size_t lds = 4 * 2 * kcap * 4;
    for (int mode = 1; mode <= 5; mode++) {
        P.ablate = mode;
        float t = timeit([&] { (void)hipMemsetAsync(cnt, 0, 16, 0);
            hipLaunchKernelGGL(k_mask_annotate<40>, dim3(2048), dim3(256), lds, 0, P, 0, nr - 1, rp, a, rl, (const int2*)nullptr, mc, kcap, mask, cmask, rf, anno, hf, aoff, acnt, cnt, 4u * nr, wl, st); });
        printf("stop after phase %d: %7.1f us   (1 histogram, 2 +mask, 3 +cov0 scan/gate, 4 +candidates, 5 all)\n", mode, t * 1e3);
    }
    P.ablate = 5;
    for (int g : {768, 1024, 1280, 1536, 1792, 2048, 2304, 2560, 3072, 3584, 4096, 6144, 8192, 21850}) {
        float t = timeit([&] { (void)hipMemsetAsync(cnt, 0, 16, 0);
            hipLaunchKernelGGL(k_mask_annotate<40>, dim3(g), dim3(256), lds, 0, P, 0, nr - 1, rp, a, rl, (const int2*)nullptr, mc, kcap, mask, cmask, rf, anno, hf, aoff, acnt, cnt, 4u * nr, wl, st); });
        printf("grid %5d: %7.1f us\n", g, t * 1e3);
    }
    return 0;
}
