// microbenchmark: HBM read bandwidth by access pattern (int2 elements)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// (a) flat grid-stride, U loads in flight
template <int U, typename T>
__global__ __launch_bounds__(256) void k_flat(const T* __restrict__ a, long n, int* out) {
    long i = (long)blockIdx.x * 256 * U + threadIdx.x;
    const long stride = (long)gridDim.x * 256 * U;
    int acc = 0;
    for (; i < n; i += stride) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { long k = i + (long)u * 256; v[u] = k < n ? a[k] : T{}; }
#pragma unroll
        for (int u = 0; u < U; u++) acc += ((int*)&v[u])[0] ^ ((int*)&v[u])[sizeof(T) / 4 - 1];
    }
    if (acc == 0x12345678) out[0] = acc;
}
// (b) one wave per segment of `seg` int2 elements, segments visited with stride nwaves (like reads)
template <int U>
__global__ __launch_bounds__(256) void k_seg(const int2* __restrict__ a, long nseg, int seg, int* out) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * 256) >> 6;
    int acc = 0;
    for (long s = wave; s < nseg; s += nwaves) {
        const long b = s * seg, e = b + seg;
        for (long base = b; base < e; base += U * 64) {
            int2 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) { long k = base + u * 64 + lane; v[u] = k < e ? a[k] : make_int2(0, 0); }
#pragma unroll
            for (int u = 0; u < U; u++) acc += v[u].x ^ v[u].y;
        }
    }
    if (acc == 0x12345678) out[0] = acc;
}
// (c) block per group of consecutive segments: flat within the block's contiguous range
template <int U>
__global__ __launch_bounds__(256) void k_blockrange(const int2* __restrict__ a, long n, long per_block, int* out) {
    int acc = 0;
    for (long blk = blockIdx.x; blk * per_block < n; blk += gridDim.x) {
        const long b = blk * per_block, e = min(n, b + per_block);
        for (long base = b; base < e; base += U * 256) {
            int2 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) { long k = base + u * 256 + threadIdx.x; v[u] = k < e ? a[k] : make_int2(0, 0); }
#pragma unroll
            for (int u = 0; u < U; u++) acc += v[u].x ^ v[u].y;
        }
    }
    if (acc == 0x12345678) out[0] = acc;
}

template <typename F>
float timeit(F f, int reps = 10) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const long n = 26220172;   // overlaps of the E. coli workload
    int2* a; int* out;
    CK(hipMalloc(&a, n * sizeof(int2))); CK(hipMalloc(&out, 4));
    CK(hipMemset(a, 1, n * sizeof(int2)));
    const double gb = n * 8 / 1e9;
    for (int grid : {2048, 4096, 8192}) {
        float t;
        t = timeit([&] { hipLaunchKernelGGL((k_flat<4, int2>), dim3(grid), dim3(256), 0, 0, a, n, out); });
        printf("flat int2  U=4 grid %5d: %7.1f us  %6.2f TB/s\n", grid, t * 1e3, gb / t);
        t = timeit([&] { hipLaunchKernelGGL((k_flat<8, int2>), dim3(grid), dim3(256), 0, 0, a, n, out); });
        printf("flat int2  U=8 grid %5d: %7.1f us  %6.2f TB/s\n", grid, t * 1e3, gb / t);
        t = timeit([&] { hipLaunchKernelGGL((k_flat<4, int4>), dim3(grid), dim3(256), 0, 0, (const int4*)a, n / 2, out); });
        printf("flat int4  U=4 grid %5d: %7.1f us  %6.2f TB/s\n", grid, t * 1e3, gb / t);
    }
    for (int seg : {300, 304, 512, 2048}) {
        for (int grid : {2048, 8192}) {
            long nseg = n / seg;
            float t = timeit([&] { hipLaunchKernelGGL((k_seg<8>), dim3(grid), dim3(256), 0, 0, a, nseg, seg, out); });
            printf("wave/segment seg=%4d grid %5d: %7.1f us  %6.2f TB/s\n", seg, grid, t * 1e3, gb / t);
        }
    }
    for (long per : {4096L, 16384L, 65536L}) {
        float t = timeit([&] { hipLaunchKernelGGL((k_blockrange<8>), dim3(2048), dim3(256), 0, 0, a, n, per, out); });
        printf("block/range per=%6ld: %7.1f us  %6.2f TB/s\n", per, t * 1e3, gb / t);
    }
    return 0;
}
