// Does the rate of v_mfma_f32_32x32x16_bf16 on a SIMD that one wave has to itself depend on WHERE its operands come from?
// 36 MFMAs per iteration (two accumulators taking turns), 1 workgroup of 256 threads per CU:
//   mode 0: one A and one B register quad for every MFMA (the usual microbenchmark)
//   mode 1: six A quads x three B quads, rotated as a nine-product bf16-split step does
//   mode 2: mode 1 with the accumulators in AGPRs
//   mode 3: mode 1, operand quads at odd register offsets (bank alignment)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define MV(acc, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B))
#define MA(acc, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(A), "v"(B))
template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(float* out, int iters, int seed, unsigned long long* cyc) {
    extern __shared__ char dyn_lds[];
    if (threadIdx.x >= 256) {                     // (512-thread launches: waves 4-7 leave at once, or idle at a barrier: seed bit 8)
        if (seed & 256) __builtin_amdgcn_s_barrier();
        return;
    }
    f32x16 c0, c1;
    for (int v = 0; v < 16; ++v) { c0[v] = 0.f; c1[v] = 0.f; }
    i32x4 a[6], b[3];
    for (int i = 0; i < 6; ++i) a[i] = i32x4{0x3f803f80 + seed * i, 0x3f803f80, 0x3f803f80, 0x3f803f80 + i};
    for (int i = 0; i < 3; ++i) b[i] = i32x4{0x3f803f80, 0x3f803f80 + seed, 0x3f803f80 + i, 0x3f803f80};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int s = 4; s >= 0; --s)
#pragma unroll
                for (int pa = 0; pa < 3; ++pa) {
                    const int pb = s - pa;
                    if (pb < 0 || pb >= 3) continue;
                    if (MODE == 0) { MV(c0, a[0], b[0]); MV(c1, a[0], b[0]); }
                    if (MODE == 1 || MODE == 3) { MV(c0, a[pb], b[pa]); MV(c1, a[3 + pb], b[pa]); }
                    if (MODE == 2) { MA(c0, a[pb], b[pa]); MA(c1, a[3 + pb], b[pa]); }
                }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (blockDim.x > 256 && (seed & 256)) __builtin_amdgcn_s_barrier();
    float s = 0; for (int v = 0; v < 16; ++v) s += c0[v] + c1[v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* what, float* out, unsigned long long* cyc, int blocks, int threads, int lds = 0, int seed = 0) {
    const int iters = 400;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), lds, 0, out, iters, seed, cyc); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), lds, 0, out, iters, seed, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024]; hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < blocks; ++i) c += h[i];
    const double n = iters * 36.0;
    printf("%-70s blocks %4d x %4d threads: %6.1f cycles per MFMA, %6.2f ns per MFMA (wall)\n", what, blocks, threads, c / blocks / n, ms * 1e6 / n);
}
int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 8192);
    run<0>("one A quad, one B quad", out, cyc, 256, 256);
    run<1>("six A quads x three B quads (a nine-product step)", out, cyc, 256, 256);
    run<2>("... accumulators in AGPRs", out, cyc, 256, 256);
    run<1>("nine-product step, 512-thread workgroups (waves 4-7 exit)", out, cyc, 256, 512);
    run<1>("nine-product step, 512 threads, waves 4-7 wait at a barrier", out, cyc, 256, 512, 0, 256);
    run<1>("nine-product step, 256 threads, 112 KB of LDS", out, cyc, 256, 256, 112 * 1024);
    run<1>("nine-product step, 512 threads (4 exit), 112 KB of LDS", out, cyc, 256, 512, 112 * 1024);
    run<1>("nine-product step, 256 threads, 512 workgroups", out, cyc, 512, 256);
    return 0;
}
