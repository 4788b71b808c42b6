// What can share a SIMD with back-to-back fp32 MFMAs?  Each variant: 2 accumulators, 8 MFMAs
// per block of fillers, pinned order via volatile inline asm.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(acc) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define VADD(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y))
#define VMUL(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(y))
#define VCND(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y) : )
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a, float b, int y) {
    __shared__ float4 lds[1024];
    f32x16 c0, c1;
    for (int v = 0; v < 16; ++v) { c0[v] = 0.f; c1[v] = 0.f; }
    int x0 = threadIdx.x, x1 = y, x2 = 3, x3 = 5;
    lds[threadIdx.x] = make_float4(a, b, a, b);
    __syncthreads();
    float4 f = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            MF(c0);
            if (MODE == 1) { VADD(x0); VADD(x1); VADD(x2); VADD(x3); VADD(x0); VADD(x1); }
            if (MODE == 2) { VMUL(x0); }
            if (MODE == 3) { VMUL(x0); VMUL(x1); VADD(x2); VADD(x3); VADD(x2); VADD(x3); }
            if (MODE == 5) { for (int q = 0; q < 12; ++q) VADD(x0); }
            MF(c1);
            if (MODE == 1) { VADD(x0); VADD(x1); VADD(x2); VADD(x3); VADD(x0); VADD(x1); }
            if (MODE == 2) { VMUL(x1); }
            if (MODE == 3) { VMUL(x0); VMUL(x1); VADD(x2); VADD(x3); VADD(x2); VADD(x3); }
            if (MODE == 5) { for (int q = 0; q < 12; ++q) VADD(x1); }
            if (MODE == 4) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(f) : "v"((threadIdx.x & 63) * 16 + r * 1024));
            }
        }
        if (MODE == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = x0 + x1 + x2 + x3 + f.x + f.y;
    for (int v = 0; v < 16; ++v) s += c0[v] + c1[v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* what, int blocks, int iters) {
    float* out; (void)hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<MODE><<<blocks, 256>>>(out, iters, 1.f, 1.f, 7);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    probe<MODE><<<blocks, 256>>>(out, iters, 1.f, 1.f, 7);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * 8 * 4096.0;
    printf("%-44s blocks=%4d: %.3f ms  %6.1f TF/s\n", what, blocks, ms, flop / ms / 1e9);
    (void)hipFree(out);
}
int main() {
    for (int blocks : {256, 512}) {
        run<0>("mfma only", blocks, 2000);
        run<1>("mfma + 6 v_add per mfma", blocks, 2000);
        run<5>("mfma + 12 v_add per mfma", blocks, 2000);
        run<2>("mfma + 1 v_mul_lo per mfma", blocks, 2000);
        run<3>("mfma + 2 v_mul_lo + 4 v_add per mfma", blocks, 2000);
        run<4>("mfma + 1 ds_read_b128 per 2 mfma", blocks, 2000);
    }
    return 0;
}
