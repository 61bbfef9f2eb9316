// What a same-address LDS atomic costs:  hipcc --offload-arch=gfx950 -O3 -o lds_atomic_probe tools/probes/lds_atomic_probe.cpp && ./lds_atomic_probe
// Every wavefront of a full GPU (8 per SIMD) issues ITER ds_add_u32 whose 64 lanes hit D distinct words (D = 64: none shared,
// 1: all the same word); reported: ns per instruction per wavefront and the aggregate rate per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int D>
__global__ __launch_bounds__(256) void k(int iters, unsigned* out) {
    __shared__ unsigned bins[4 * 1024];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int t = threadIdx.x; t < 4096; t += 256) bins[t] = 0;
    __syncthreads();
    unsigned* base = bins + w * 1024;
    const int idx = (lane % D) * (64 / D == 0 ? 1 : 1);   // D distinct words, lanes l and l + D share one
    for (int it = 0; it < iters; it++) {
        __hip_atomic_fetch_add(base + ((idx + it * 7) & 1023), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = bins[0];
}
template <int D> void run(int iters) {
    unsigned* out; hipMalloc(&out, 4 * 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * 8;
    k<D><<<grid, 256>>>(iters, out);
    hipDeviceSynchronize();
    hipEventRecord(a); k<D><<<grid, 256>>>(iters, out); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // 8 waves per SIMD resident, 32 per CU
    printf("D=%2d distinct words: %.1f us, %.1f ns per ds_add per wavefront, %.2f ns per ds_add per CU (32 wavefronts)\n", D, ms * 1e3, ms * 1e6 / iters, ms * 1e6 / iters / 32);
    hipFree(out);
}
int main() { const int it = 20000; run<64>(it); run<32>(it); run<16>(it); run<8>(it); run<4>(it); run<2>(it); run<1>(it); return 0; }
