// How long does a wave wait for its kernel arguments?  Cycles from the wave's first instruction until a 384-byte by-value
// argument struct has arrived in SGPRs (s_load from the kernarg segment), for plain launches and hipGraph replays.
// Run with HIP_FORCE_DEV_KERNARG=0 / 1 in the environment.   usage: kernarg_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
struct Big { int v[96]; };
__global__ __launch_bounds__(256) void probe(const Big a, unsigned long long* out) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    __builtin_amdgcn_sched_barrier(0);
    int s = 0;
#pragma unroll
    for (int i = 0; i < 96; ++i) s += a.v[i];
    asm volatile("" :: "s"(s));
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = s; }
}
int main() {
    unsigned long long* d; CK(hipMalloc(&d, 1024 * 16));
    Big a; for (int i = 0; i < 96; ++i) a.v[i] = i;
    hipStream_t st; CK(hipStreamCreate(&st));
    auto report = [&](const char* what) {
        std::vector<unsigned long long> h(1024 * 2);
        CK(hipMemcpy(h.data(), d, 1024 * 16, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> c; for (int i = 0; i < 432; ++i) c.push_back(h[2 * i]);
        std::sort(c.begin(), c.end());
        printf("%-40s cycles until the arguments are in registers: p10 %llu, median %llu, p90 %llu\n", what, c[43], c[216], c[388]);
    };
    for (int r = 0; r < 3; ++r) { a.v[0] = r; hipLaunchKernelGGL(probe, dim3(432), dim3(256), 0, st, a, d); CK(hipStreamSynchronize(st)); }
    report("plain launch");
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int r = 0; r < 10; ++r) { a.v[0] = r; hipLaunchKernelGGL(probe, dim3(432), dim3(256), 0, st, a, d); }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) { CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); }
    report("hipGraph replay (10 launches, the last)");
    return 0;
}
