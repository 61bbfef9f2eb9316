// which ingredient of k_cov_stats costs the time?  (uniform rows of 300 overlaps)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <limits.h>
#include "../hinge_amd/csrc/filter_kernels.h"
using namespace hinge;
template <int MODE>
__global__ __launch_bounds__(256) void k_var(int r_begin, int r_end, const int64_t* __restrict__ row_ptr, const int2* __restrict__ a_span,
                                             const int* __restrict__ rlen, int* __restrict__ mean_cov, int* __restrict__ nbins0,
                                             unsigned long long* __restrict__ totals) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * 256) >> 6;
    long long blk = 0;
    for (int i = r_begin + wave; i <= r_end; i += nwaves) {
        int64_t s, e;
        if (MODE >= 1) { s = row_ptr[i]; e = row_ptr[i + 1]; } else { s = (int64_t)i * 300; e = s + 300; }
        int rl = MODE >= 1 ? rlen[i] : 9000;
        int sum = 0, mx = INT_MIN;
        for (int64_t base = s; base < e; base += 8 * 64) {
            int2 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int64_t k = base + u * 64 + lane; v[u] = k < e ? a_span[k] : make_int2(0, 0); }
#pragma unroll
            for (int u = 0; u < 8; u++) if (base + u * 64 + lane < e) { sum += bin_of<40>(v[u].y, 40) - bin_of<40>(v[u].x, 40); mx = max(mx, max(v[u].x, v[u].y)); }
        }
        if (MODE >= 2) {
            const long long tot = wave_sum64((long long)sum);
            mx = wave_max(mx);
            if (lane == 0) {
                const int K = nbins_of<40>((int)(e - s), mx, 40);
                if (MODE >= 3) nbins0[i] = K;
                if (rl >= 5000) {
                    const long long m = MODE >= 4 ? tot / (long long)max(1, K) : tot;
                    if (MODE >= 3) mean_cov[i] = (int)m;
                    blk += tot + K;
                } else if (MODE >= 3) mean_cov[i] = INT_MIN;
            }
        } else blk += sum + mx;
    }
    if (lane == 0 && blk == 0x123456789LL) atomicAdd(&totals[0], (unsigned long long)blk);
}
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
    const int nr = 87400; const long n = (long)nr * 300;
    int2* a; int64_t* rp; int* rl; int* mc; int* nb; unsigned long long* tot;
    (void)hipMalloc(&a, n * 8); (void)hipMalloc(&rp, (nr + 1) * 8); (void)hipMalloc(&rl, nr * 4); (void)hipMalloc(&mc, nr * 4); (void)hipMalloc(&nb, nr * 4); (void)hipMalloc(&tot, 16);
    std::vector<int64_t> h(nr + 1); for (int i = 0; i <= nr; i++) h[i] = (int64_t)i * 300;
    std::vector<int> hl(nr, 9000);
    std::vector<int2> ha(n); for (long i = 0; i < n; i++) ha[i] = make_int2((int)(i % 5000), (int)(i % 5000) + 2000);
    (void)hipMemcpy(rp, h.data(), (nr + 1) * 8, hipMemcpyHostToDevice); (void)hipMemcpy(rl, hl.data(), nr * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(a, ha.data(), n * 8, hipMemcpyHostToDevice);
    const double gb = n * 8 / 1e9;
#define RUN(M) { float t = timeit([&] { hipLaunchKernelGGL((k_var<M>), dim3(2048), dim3(256), 0, 0, 0, nr - 1, rp, a, rl, mc, nb, tot); }); printf("mode %d: %7.1f us %6.2f TB/s\n", M, t * 1e3, gb / t); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
    { float t = timeit([&] { hipLaunchKernelGGL((k_cov_stats<40>), dim3(2048), dim3(256), 0, 0, 0, nr - 1, rp, a, rl, 40, mc, nb, tot); }); printf("k_cov_stats: %7.1f us %6.2f TB/s\n", t * 1e3, gb / t); }
    return 0;
}
