// What does a split-bf16 contraction cost next to the fp32 MFMA?  v_mfma_f32_32x32x16_bf16 back to back, then with
// the vector work a three-way bf16 split of fp32 operands needs (and / sub / perm) and with the LDS fragment reads.
// Rates are printed as "fp32-equivalent" TF/s for 6 and 9 bf16 products per fp32 product.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define MFB(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define MFBV(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MF32(acc) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(fa), "v"(fb))
#define VADD(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y))
#define VAND(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(y))
#define VSUBF(x) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(fy))
#define VPERM(x) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(sel))
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float fa, float fb, int y, const float4* src) {
    __shared__ float4 lds[2048];
    f32x16 c0, c1, c2, c3;
    for (int v = 0; v < 16; ++v) { c0[v] = 0.f; c1[v] = 0.f; c2[v] = 0.f; c3[v] = 0.f; }
    i32x4 a = {0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80}, b = a;
    int x0 = threadIdx.x, x1 = y, x2 = 3, x3 = 5, sel = 0x07060302;
    float fx0 = fa, fx1 = fb, fy = 0.25f;
    lds[threadIdx.x] = make_float4(fa, fb, fa, fb);
    __syncthreads();
    float4 f0 = make_float4(0, 0, 0, 0), f1 = f0, f2 = f0;
    // vector-memory modes: a wave's load is 1 KB contiguous (14, 15: every 4 / 8 MFMAs), one 128-byte line per lane (16), or an
    // LDS-DMA of 1 KB (17); the source is a 1 MB buffer that stays in L2
    const float4* gsrc = src + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & 255) * 256 + (MODE == 16 ? (threadIdx.x & 63) * 8 : (MODE == 18 || MODE == 19) ? (threadIdx.x & 31) * 8 + ((threadIdx.x >> 5) & 1) : (threadIdx.x & 63));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (MODE == 9) { MF32(c0); MF32(c1); MF32(c2); MF32(c3); continue; }
            if (MODE == 8) { MFBV(c0); MFBV(c1); MFBV(c2); MFBV(c3); continue; }
            // dependent chains: ONE accumulator (the 128x32-tile kernels: a wave owns one 32x32 tile), two taking turns
            if (MODE == 10) { MFBV(c0); MFBV(c0); MFBV(c0); MFBV(c0); continue; }
            if (MODE == 11) { MFBV(c0); MFBV(c1); MFBV(c0); MFBV(c1); continue; }
            if (MODE == 12) { MFBV(c0); VAND(x0); VSUBF(fx0); VPERM(x1); VAND(x2); MFBV(c0); VAND(x0); VSUBF(fx0); VPERM(x1); VAND(x2);
                              MFBV(c0); VAND(x0); VSUBF(fx0); VPERM(x1); VAND(x2); MFBV(c0); VAND(x0); VSUBF(fx0); VPERM(x1); VAND(x2); continue; }
            if (MODE == 13) { MFBV(c0); VAND(x0); VSUBF(fx0); VPERM(x1); VAND(x2); MFBV(c1); VAND(x0); VSUBF(fx0); VPERM(x1); VAND(x2);
                              MFBV(c0); VAND(x0); VSUBF(fx0); VPERM(x1); VAND(x2); MFBV(c1); VAND(x0); VSUBF(fx0); VPERM(x1); VAND(x2); continue; }
            MFB(c0);
            if (MODE == 1) { VADD(x0); VADD(x1); VADD(x2); VADD(x3); }
            if (MODE == 2) { VADD(x0); VADD(x1); VADD(x2); VADD(x3); VADD(x0); VADD(x1); VADD(x2); VADD(x3); }
            if (MODE == 3) { VAND(x0); VSUBF(fx0); VPERM(x1); VAND(x2); VSUBF(fx1); VPERM(x3); }
            MFB(c1);
            if (MODE == 1) { VADD(x0); VADD(x1); VADD(x2); VADD(x3); }
            if (MODE == 2) { VADD(x0); VADD(x1); VADD(x2); VADD(x3); VADD(x0); VADD(x1); VADD(x2); VADD(x3); }
            if (MODE == 3) { VAND(x0); VSUBF(fx0); VPERM(x1); VAND(x2); VSUBF(fx1); VPERM(x3); }
            if (MODE == 4 || MODE == 5) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(f0) : "v"((threadIdx.x & 63) * 16 + r * 1024));
            }
            if (MODE == 14 || (MODE == 15 && r == 0) || MODE == 16 || MODE == 18 || (MODE == 19 && r == 0)) {
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(f0) : "v"(gsrc + (MODE == 16 || MODE == 18 || MODE == 19 ? 2 * r : 64 * r)) : "memory");
            }
            if (MODE == 17) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + 64 * r),
                                                 (__attribute__((address_space(3))) void*)(lds + (threadIdx.x >> 6) * 64 + 1024), 16, 0, 0);
            }
            MFB(c2);
            if (MODE == 1) { VADD(x0); VADD(x1); VADD(x2); VADD(x3); }
            if (MODE == 2) { VADD(x0); VADD(x1); VADD(x2); VADD(x3); VADD(x0); VADD(x1); VADD(x2); VADD(x3); }
            if (MODE == 3) { VAND(x0); VSUBF(fx0); VPERM(x1); VAND(x2); VSUBF(fx1); VPERM(x3); }
            if (MODE == 5) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(f1) : "v"((threadIdx.x & 63) * 16 + r * 1024 + 4096));
            }
            MFB(c3);
            if (MODE == 1) { VADD(x0); VADD(x1); VADD(x2); VADD(x3); }
            if (MODE == 2) { VADD(x0); VADD(x1); VADD(x2); VADD(x3); VADD(x0); VADD(x1); VADD(x2); VADD(x3); }
            if (MODE == 3) { VAND(x0); VSUBF(fx0); VPERM(x1); VAND(x2); VSUBF(fx1); VPERM(x3); }
            if (MODE == 5) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(f2) : "v"((threadIdx.x & 63) * 16 + r * 1024 + 8192));
            }
        }
        if (MODE == 4 || MODE == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (MODE >= 14 && MODE <= 19) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float s = x0 + x1 + x2 + x3 + f0.x + f1.y + f2.z + fx0 + fx1;
    for (int v = 0; v < 16; ++v) s += c0[v] + c1[v] + c2[v] + c3[v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* what, int blocks, int iters) {
    float* out; (void)hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    static float4* src = nullptr;
    if (!src) { (void)hipMalloc(&src, 2 << 20); (void)hipMemset(src, 0, 2 << 20); }
    probe<MODE><<<blocks, 256>>>(out, iters, 1.f, 1.f, 7, src);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    probe<MODE><<<blocks, 256>>>(out, iters, 1.f, 1.f, 7, src);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double macs_per_mfma = MODE == 9 ? 2048.0 : 16384.0;
    double flop = (double)blocks * 4 * iters * 8 * macs_per_mfma * 2;
    double tf = flop / ms / 1e9;
    if (MODE == 9) printf("%-52s blocks=%4d: %.3f ms  %7.1f TF/s fp32\n", what, blocks, ms, tf);
    else printf("%-52s blocks=%4d: %.3f ms  %7.1f TF/s bf16 = %6.1f (x6) / %6.1f (x9) fp32-equivalent\n", what, blocks,
                ms, tf, tf / 6, tf / 9);
    (void)hipFree(out);
}
int main() {
    for (int blocks : {256, 512, 1024}) {
        run<9>("fp32 mfma 32x32x2 only", blocks, 2000);
        run<0>("bf16 mfma 32x32x16 only (acc in AGPRs)", blocks, 2000);
        run<8>("bf16 mfma only (acc in VGPRs)", blocks, 2000);
        run<1>("bf16 mfma + 4 v_add per mfma", blocks, 2000);
        run<2>("bf16 mfma + 8 v_add per mfma", blocks, 2000);
        run<3>("bf16 mfma + (2 and, 2 sub_f32, 2 perm) per mfma", blocks, 2000);
        run<4>("bf16 mfma + 1 ds_read_b128 per 4 mfma", blocks, 2000);
        run<5>("bf16 mfma + 3 ds_read_b128 per 4 mfma", blocks, 2000);
        run<10>("bf16 mfma, ONE accumulator (dependent chain)", blocks, 2000);
        run<11>("bf16 mfma, two accumulators taking turns", blocks, 2000);
        run<12>("one accumulator + 4 vector instructions per mfma", blocks, 2000);
        run<13>("two accumulators + 4 vector instructions per mfma", blocks, 2000);
        run<14>("bf16 mfma + 1 global_load_dwordx4 (1 KB row) per 4 mfma", blocks, 2000);
        run<15>("bf16 mfma + 1 global_load_dwordx4 (1 KB row) per 8 mfma", blocks, 2000);
        run<16>("bf16 mfma + 1 global_load_dwordx4 (line per lane) per 4", blocks, 2000);
        run<17>("bf16 mfma + 1 LDS-DMA of 1 KB per 4 mfma", blocks, 2000);
        run<18>("bf16 mfma + 1 load, lane = (row, half) (32 lines) per 4", blocks, 2000);
        run<19>("bf16 mfma + 1 load, lane = (row, half) (32 lines) per 8", blocks, 2000);
    }
    return 0;
}
