// Can a lone wave per SIMD keep the nine-product MFMA stream fed when the WEIGHT fragments come straight from global memory
// (pre-split bf16 planes in fragment order, every workgroup of the chip reading the same 192 KB) instead of through LDS
// behind a per-k-tile barrier?  The image-stationary forward considered in round 6 (activations resident in LDS as bf16
// planes, no barrier in the k-loop) stands or falls with this.  Per 16-k step and wave: 3 ds_read_b128 (activation
// planes), 6 x 16-byte global loads per lane (two 32-column tiles x three planes, 6 KB per wave), 18 MFMAs on two
// accumulators; the global loads run D steps ahead in a register ring.
//   usage: bdirect_probe            (prints cycles per 36 MFMAs; 1154 = the matrix pipe saturated)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define MV(acc, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B))
constexpr int STEPS = 32;            // conv 2: 16 taps x 2 half-taps of 16 k

// MODE 0: B from global, every wave of the chip the same addresses; 1: B from LDS (no global traffic: the ceiling);
// 2: B from global, each WORKGROUP its own copy (no sharing between CUs: L2 / fabric bound)
template <int D, int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void probe(const i32x4* __restrict__ wfrag, float* out, int images, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    f32x16 c0, c1;
    for (int v = 0; v < 16; ++v) { c0[v] = 0.f; c1[v] = 0.f; }
    for (int i = threadIdx.x; i < 4096; i += WAVES * 64) reinterpret_cast<i32x4*>(lds)[i] = i32x4{0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80};
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* la = lds + lane * 16 + (wave & 3) * 4096;
    const i32x4* gw = wfrag + lane + (MODE == 2 ? (size_t)blockIdx.x * STEPS * 6 * 64 : 0);
    i32x4 ring[D][6];
    auto issue = [&](int slot, int step) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (MODE == 1) ring[slot][i] = *reinterpret_cast<const i32x4*>(lds + 16384 + ((step * 6 + i) & 31) * 1024 + lane * 16);
            else ring[slot][i] = gw[(step * 6 + i) * 64];
        }
    };
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int img = 0; img < images; ++img) {
#pragma unroll
        for (int d = 0; d < D; ++d) issue(d, d);
        for (int s0 = 0; s0 < STEPS; s0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int s = s0 + d;
                i32x4 a[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[pl] = *reinterpret_cast<const i32x4*>(la + ((s * 3 + pl) & 3) * 1024);
#pragma unroll
                for (int sum = 4; sum >= 0; --sum)
#pragma unroll
                    for (int pa = 0; pa < 3; ++pa) {
                        const int pb = sum - pa;
                        if (pb < 0 || pb >= 3) continue;
                        MV(c0, ring[d][pb], a[pa]);
                        MV(c1, ring[d][3 + pb], a[pa]);
                    }
                if (s + D < STEPS) issue(d, s + D);          // (uniform) the slot just consumed takes step s + D
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sacc = 0; for (int v = 0; v < 16; ++v) sacc += c0[v] + c1[v];
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = sacc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int D, int MODE, int WAVES> void run(const char* what, const i32x4* w, float* out, unsigned long long* cyc) {
    const int images = 40, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((probe<D, MODE, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, w, out, images, cyc); hipDeviceSynchronize(); }
    unsigned long long h[256]; hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double c = 0, mx = 0; for (int i = 0; i < blocks; ++i) { c += h[i]; if (h[i] > mx) mx = h[i]; }
    printf("%-84s %6.0f cycles per 36 MFMAs (slowest workgroup %6.0f)\n", what, c / blocks / (images * STEPS / 2.0), mx / (images * STEPS / 2.0));
}
int main() {
    i32x4* w; float* out; unsigned long long* cyc;
    const size_t n = (size_t)257 * STEPS * 6 * 64;
    hipMalloc(&w, n * 16); hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 8192);
    hipMemset(w, 0x3f, n * 16);
    printf("one workgroup per CU, one wave per SIMD (4 waves), 18 MFMAs + 3 ds_read_b128 + 6 x 16 B per lane of weights per step\n");
    run<4, 1, 4>("weights from LDS (ceiling: no global traffic)", w, out, cyc);
    run<1, 0, 4>("weights from global, shared by every workgroup, 1 step ahead", w, out, cyc);
    run<2, 0, 4>("weights from global, shared, 2 steps ahead", w, out, cyc);
    run<4, 0, 4>("weights from global, shared, 4 steps ahead", w, out, cyc);
    run<8, 0, 4>("weights from global, shared, 8 steps ahead", w, out, cyc);
    run<4, 2, 4>("weights from global, a private copy per workgroup, 4 steps ahead", w, out, cyc);
    run<8, 2, 4>("weights from global, a private copy per workgroup, 8 steps ahead", w, out, cyc);
    printf("two waves per SIMD (8 waves: each pair of waves owns one 32-row tile's two column halves -- here simply twice the waves)\n");
    run<4, 1, 8>("weights from LDS", w, out, cyc);
    run<4, 0, 8>("weights from global, shared, 4 steps ahead", w, out, cyc);
    run<8, 0, 8>("weights from global, shared, 8 steps ahead", w, out, cyc);
    return 0;
}
