// Where a fresh process's HIP start-up goes, call by call (what hinge_ctx_create does, plus the alternatives).
//   hipcc -O2 -o /tmp/hip_init_probe tools/probes/hip_init_probe.cpp -ldl && /tmp/hip_init_probe [path/to/libhinge_hip.so]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define T(what, call) do { double t0 = now(); auto r = (call); double t1 = now(); printf("%-44s %8.1f ms  (rc %d)\n", what, t1 - t0, (int)r); } while (0)
__global__ void k_touch(int* p) { p[0] = 1; }
int main(int argc, char** argv) {
    double t00 = now();
    if (argc > 1) { double t0 = now(); void* h = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL); printf("%-44s %8.1f ms  (%s)\n", "dlopen(libhinge_hip.so)", now() - t0, h ? "ok" : dlerror()); }
    int n = 0;
    T("hipInit(0)", hipInit(0));
    if (getenv("PROBE_INIT_ONLY")) { printf("%-44s %8.1f ms\n", "total", now() - t00); fflush(stdout); _exit(0); }
    T("hipGetDeviceCount", hipGetDeviceCount(&n));
    T("hipSetDevice(0)", hipSetDevice(0));
    int cu = 0;
    T("hipDeviceGetAttribute(multiprocessorCount)", hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, 0));
    hipDeviceProp_t prop;
    T("hipGetDeviceProperties", hipGetDeviceProperties(&prop, 0));
    int* p = nullptr;
    T("hipMalloc(64 B)", hipMalloc(&p, 64));
    T("hipMemset(64 B)", hipMemset(p, 0, 64));
    hipEvent_t e;
    T("hipEventCreate", hipEventCreate(&e));
    hipStream_t s;
    T("hipStreamCreate", hipStreamCreate(&s));
    { double t0 = now(); hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, s, p); hipStreamSynchronize(s); printf("%-44s %8.1f ms\n", "first kernel launch + sync", now() - t0); }
    int* big = nullptr;
    const char* mb = getenv("PROBE_ALLOC_MB");
    const size_t big_mb = mb ? (size_t)atol(mb) : 600;
    if (big_mb) { T("hipMalloc(PROBE_ALLOC_MB, default 600)", hipMalloc(&big, big_mb << 20)); }
    if (big_mb && getenv("PROBE_TOUCH")) { T("hipMemset(all of it)", hipMemset(big, 1, big_mb << 20)); T("sync", hipDeviceSynchronize()); }
    if (getenv("PROBE_FREE")) { T("hipFree(big)", hipFree(big)); T("hipFree(p)", hipFree(p)); T("hipStreamDestroy", hipStreamDestroy(s)); }
    if (getenv("PROBE_RESET")) { T("hipDeviceReset", hipDeviceReset()); }
    printf("%-44s %8.1f ms   (%d CUs, %d devices)\n", "total", now() - t00, cu, n);
    fflush(stdout);
    if (getenv("PROBE_EARLY_SHUTDOWN")) {   // give the device back BEFORE the (simulated) text output: does the next process start on a quiet GPU?
        typedef int (*shut_t)(void);
        shut_t shut = (shut_t)dlsym(RTLD_DEFAULT, "hsa_shut_down");
        double t0 = now();
        int rc = -1, rounds = 0;
        if (getenv("PROBE_RESET_FIRST")) (void)hipDeviceReset();
        while (shut && rounds < 8) { rc = shut(); rounds++; if (rc != 0) break; }   // (reference counted: until it says "not initialised")
        printf("%-44s %8.1f ms  (%d calls, last rc %d)\n", "hsa_shut_down", now() - t0, rounds, rc);
        fflush(stdout);
        usleep(1000 * atoi(getenv("PROBE_EARLY_SHUTDOWN")));
        _exit(0);
    }
    if (getenv("PROBE_SLEEP_MS")) usleep(1000 * atoi(getenv("PROBE_SLEEP_MS")));
    if (getenv("PROBE_FAST_EXIT")) _exit(0);   // what the executables do: no runtime teardown of its own, the kernel's only
    return 0;
}
