// short launches of pure dependent-chain fp32 MFMAs: what does a 25-30 us kernel of NOTHING BUT MFMAs reach?
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
    if (s == 12345.f) out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int blocks, int iters, int launches) {
    float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int k = 0; k < 3; ++k) probe<NACC><<<blocks, 256>>>(out, iters, 1.f, 1.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int k = 0; k < launches; ++k) probe<NACC><<<blocks, 256>>>(out, iters, 1.f, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)launches * blocks * 4 * iters * 8 * NACC * 4096.0;
    printf("NACC=%d blocks=%4d mfma/wave=%5d launches=%d: %.1f us per launch  %.1f TF/s\n", NACC, blocks, iters * 8 * NACC, launches, ms * 1e3 / launches, flop / ms / 1e9);
    (void)hipFree(out);
}
int main() {
    // conv 1 forward's shape: 1900 tiles x 128 MFMAs per wave
    run<1>(1900, 16, 20);      // one tile per workgroup
    run<1>(1024, 30, 20);      // ~ the persistent split (1024 x 240 MFMAs ~ 1900 x 128)
    run<1>(256, 119, 20);      // one workgroup per CU
    run<1>(512, 59, 20);
    run<2>(512, 30, 20);
    run<1>(1024, 300, 5);      // long
    run<1>(864, 32, 20);       // conv 2 forward, 64x64 tiles: 864 x 256
    run<1>(1024, 27, 20);
    return 0;
}
