// where does HIP start-up time go?  (hipInit, device, stream, first allocation, first launch = code object load)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void k(int* p) { if (p) *p = 1; }
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    double t0 = now(), t;
    (void)hipInit(0); t = now(); printf("hipInit            %7.1f ms\n", t - t0); t0 = t;
    int n = 0; (void)hipGetDeviceCount(&n); t = now(); printf("hipGetDeviceCount  %7.1f ms (%d)\n", t - t0, n); t0 = t;
    (void)hipSetDevice(0); t = now(); printf("hipSetDevice       %7.1f ms\n", t - t0); t0 = t;
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0); t = now(); printf("hipGetDeviceProps  %7.1f ms\n", t - t0); t0 = t;
    hipStream_t s = nullptr;
    if (getenv("PROBE_STREAM")) { (void)hipStreamCreate(&s); t = now(); printf("hipStreamCreate    %7.1f ms\n", t - t0); t0 = t; }
    if (getenv("PROBE_NONBLOCK")) { (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); t = now(); printf("hipStreamCreate NB %7.1f ms\n", t - t0); t0 = t; }
    int* d; (void)hipMalloc(&d, 4096); t = now(); printf("hipMalloc (first)  %7.1f ms\n", t - t0); t0 = t;
    hipEvent_t e; (void)hipEventCreate(&e); t = now(); printf("hipEventCreate     %7.1f ms\n", t - t0); t0 = t;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d); (void)hipStreamSynchronize(s); t = now(); printf("first launch+sync  %7.1f ms\n", t - t0); t0 = t;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d); (void)hipStreamSynchronize(s); t = now(); printf("second launch+sync %7.1f ms\n", t - t0); t0 = t;
    void* big; (void)hipMalloc(&big, 600u << 20); t = now(); printf("hipMalloc 600 MB   %7.1f ms\n", t - t0); t0 = t;
    return 0;
}
