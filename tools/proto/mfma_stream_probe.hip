// Which ingredient of a real k-loop slows a lone wave's MFMA stream?  Per iteration: 2 steps x 18 MFMAs (two accumulators
// taking turns, operands rotating over 6 + 3 quads of the step's register set), optionally with
//   bit 0: 3 vector instructions per MFMA (and / sub / perm) that WRITE the other set's B quads
//   bit 1: 8 ds_read_b128 per step that WRITE the other set's A quads (and two spare quads)
//   bit 2: s_nop 0 after every MFMA
//   bit 3: the accumulators' MFMAs in pairs of the SAME accumulator (acc0, acc0, acc1, acc1)
//   bit 4: an s_barrier per iteration
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define MV(acc, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B))
template <int F>
__global__ __launch_bounds__(256) void probe(float* out, int iters, int seed, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    f32x16 c0, c1;
    for (int v = 0; v < 16; ++v) { c0[v] = 0.f; c1[v] = 0.f; }
    i32x4 a[2][6], b[2][3], xr[2];
    for (int s = 0; s < 2; ++s) {
        for (int i = 0; i < 6; ++i) a[s][i] = i32x4{0x3f803f80 + seed * i, 0x3f803f80, 0x3f803f80 + s, 0x3f803f80 + i};
        for (int i = 0; i < 3; ++i) b[s][i] = i32x4{0x3f803f80, 0x3f803f80 + seed, 0x3f803f80 + i, 0x3f803f80 + s};
    }
    xr[0] = xr[1] = i32x4{1, 2, 3, 4};
    for (int i = threadIdx.x; i < 2048; i += 256) reinterpret_cast<i32x4*>(lds)[i] = i32x4{0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80};
    __syncthreads();
    const char* lp = lds + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 8192;
    int y = seed | 0xffff0000;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int o = st ^ 1;
            if (F & 2) {
                xr[0] = *reinterpret_cast<const i32x4*>(lp); xr[1] = *reinterpret_cast<const i32x4*>(lp + 1024);
#pragma unroll
                for (int i = 0; i < 6; ++i) a[o][i] = *reinterpret_cast<const i32x4*>(lp + 2048 + i * 1024);
            }
#pragma unroll
            for (int s = 4, n = 0; s >= 0; --s)
#pragma unroll
                for (int pa = 0; pa < 3; ++pa) {
                    const int pb = s - pa;
                    if (pb < 0 || pb >= 3) continue;
#pragma unroll
                    for (int j = 0; j < 2; ++j, ++n) {
                        if (F & 8) { if ((n >> 1) & 1) MV(c1, a[st][3 * (n & 1) + pb], b[st][pa]); else MV(c0, a[st][3 * (n & 1) + pb], b[st][pa]); }
                        else { if (j) MV(c1, a[st][3 + pb], b[st][pa]); else MV(c0, a[st][pb], b[st][pa]); }
                        if (F & 4) asm volatile("s_nop 0");
                        if ((F & 1) && n >= 2) {
                            const int q = (n - 2) >> 2, e = (n - 2) & 3;          // 16 slices -> the 12 dwords of b[o]
                            int d = b[o][q % 3][e];
                            asm volatile("v_and_b32 %0, %0, %1\n\tv_sub_f32 %0, %0, %2\n\tv_perm_b32 %0, %0, %1, %3" : "+v"(d) : "v"(y), "v"(xr[q & 1][e]), "v"(0x07060302));
                            b[o][q % 3][e] = d;
                        }
                    }
                }
        }
        if (F & 16) __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int v = 0; v < 16; ++v) s += c0[v] + c1[v];
    out[blockIdx.x * 256 + threadIdx.x] = s + b[0][0][0] + b[1][0][0] + a[0][0][0] + a[1][0][0];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int F> void run(const char* what, float* out, unsigned long long* cyc) {
    const int iters = 400, blocks = 256;
    hipLaunchKernelGGL(probe<F>, dim3(blocks), dim3(256), 0, 0, out, iters, 0, cyc); hipDeviceSynchronize();
    hipLaunchKernelGGL(probe<F>, dim3(blocks), dim3(256), 0, 0, out, iters, 0, cyc); hipDeviceSynchronize();
    unsigned long long h[256]; hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < blocks; ++i) c += h[i];
    printf("%-72s %6.1f cycles per MFMA = %5.0f per 36\n", what, c / blocks / (iters * 36.0), c / blocks / iters);
}
int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 8192);
    run<0>("MFMAs only", out, cyc);
    run<4>("+ s_nop 0 after every MFMA", out, cyc);
    run<1>("+ 3 vector instructions per MFMA (writing the other set)", out, cyc);
    run<2>("+ 8 ds_read_b128 per step (writing the other set)", out, cyc);
    run<3>("+ both", out, cyc);
    run<7>("+ both + s_nop", out, cyc);
    run<23>("+ both + s_nop + barrier", out, cyc);
    run<8>("MFMAs only, same accumulator twice in a row", out, cyc);
    run<11>("same accumulator twice in a row + both", out, cyc);
    return 0;
}
