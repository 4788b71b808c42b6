// fp32 MFMA issue-rate probe: NACC independent 32x32x2 accumulators per wave, no memory traffic.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int v = 0; v < 16; ++v) s += acc[i][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the same with operands that differ per lane and per instruction (random data toggles the datapath: the
// sustained clock under power management, and with it the real ceiling, is lower than with constants)
template <int NACC>
__global__ __launch_bounds__(256) void probe_rand(float* out, int iters, unsigned seed) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    float ar[8], br[8];
    unsigned h = seed ^ (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    for (int r = 0; r < 8; ++r) {
        h = h * 1664525u + 1013904223u; ar[r] = ((h >> 8) & 0xffff) * (1.f / 65536.f) - 0.5f;
        h = h * 1664525u + 1013904223u; br[r] = ((h >> 8) & 0xffff) * (1.f / 65536.f) - 0.5f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[r], br[(r + i) & 7], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int v = 0; v < 16; ++v) s += acc[i][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run_rand(int blocks, int iters) {
    float* out; hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe_rand<NACC><<<blocks, 256>>>(out, iters, 7u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int k = 0; k < 4; ++k) probe_rand<NACC><<<blocks, 256>>>(out, iters, 11u + k);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = 4.0 * blocks * 4 * iters * 8 * NACC * 4096.0;
    printf("random operands NACC=%d blocks=%d iters=%d: %.3f ms  %.1f TF/s\n", NACC, blocks, iters, ms, flop / ms / 1e9);
    hipFree(out);
}
template <int NACC> void run(int blocks, int iters) {
    float* out; hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NACC><<<blocks, 256>>>(out, iters, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NACC><<<blocks, 256>>>(out, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * 8 * NACC * 4096.0;
    printf("NACC=%d blocks=%d iters=%d: %.3f ms  %.1f TF/s\n", NACC, blocks, iters, ms, flop / ms / 1e9);
    hipFree(out);
}
int main() {
    run<1>(256, 2000); run<2>(256, 1000); run<4>(256, 500);
    run<2>(512, 1000); run<2>(432, 1000); run<2>(1024, 500); run<2>(216, 1000);
    run_rand<2>(256, 20000); run_rand<2>(512, 10000); run_rand<4>(256, 10000); run_rand<2>(1024, 5000);
    run<2>(512, 20000);
    return 0;
}
