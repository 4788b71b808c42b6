// What one CU can pull out of L2: bytes per cycle of LDS-DMA (buffer_load ... lds) and of plain 16-byte loads, every CU
// streaming a region of its OWN (L2-resident after the first pass) or all CUs the SAME region, by loader waves per CU
// and 1 KB pieces kept in flight per wave.  The contraction kernels' k-tiles arrive through exactly this path.
// usage: cu_bw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// each workgroup: WAVES loader waves; wave w streams its region round and round: `pieces` 1 KB pieces per pass, DEPTH in
// flight; region_bytes per workgroup; shared != 0: every workgroup reads region 0
template <int WAVES, int DEPTH, bool DMA>
__global__ __launch_bounds__(WAVES * 64) void probe(const char* src, unsigned region_bytes, int iters, int shared, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (size_t)(shared ? 0 : blockIdx.x) * region_bytes;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, region_bytes, 0x00020000);
    const unsigned per_wave = region_bytes / WAVES;
    unsigned off = wave * per_wave;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const unsigned o = wave * per_wave + (off - wave * per_wave + d * 1024) % per_wave + lane * 16;
            if constexpr (DMA) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds + (wave * DEPTH + d) * 1024), 16, o, 0, 0, 0);
            } else {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0);
                acc += v;
            }
        }
        off += DEPTH * 1024;
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (!DMA && acc.x == 0x12345678u) sink[0] = 1.f;
    if (DMA && iters < 0) sink[0] = *reinterpret_cast<float*>(lds + lane * 4);
}

template <int WAVES, int DEPTH, bool DMA>
void run(const char* src, unsigned region, int shared, float* sink) {
    auto k = probe<WAVES, DEPTH, DMA>;
    const int iters = 200;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(256), dim3(WAVES * 64), WAVES * DEPTH * 1024, 0, src, region, iters, shared, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(256), dim3(WAVES * 64), WAVES * DEPTH * 1024, 0, src, region, iters, shared, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 256.0 * WAVES * DEPTH * 1024.0 * iters;
    printf("%s %d waves x %2d KB in flight each, region %4u KB %s: %6.1f GB/s per CU = %5.1f B/clk at 2.0 GHz, chip %5.2f TB/s\n",
           DMA ? "LDS-DMA " : "reg load", WAVES, DEPTH, region / 1024, shared ? "(ONE region for all CUs)" : "(own region per CU)   ",
           bytes / 256 / (ms * 1e-3) * 1e-9, bytes / 256 / (ms * 1e-3) / 2.0e9, bytes / (ms * 1e-3) * 1e-12);
}

int main() {
    const unsigned region = 96 * 1024;                 // per CU: 256 x 96 KB = 24 MB in all: 3 MB per XCD, L2-resident
    char* src; float* sink;
    CK(hipMalloc(&src, (size_t)256 * 1024 * 1024)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(src, 1, (size_t)256 * 1024 * 1024));
    for (int shared = 0; shared < 2; ++shared) {
        run<1, 4, true>(src, region, shared, sink);  run<1, 8, true>(src, region, shared, sink);  run<1, 16, true>(src, region, shared, sink);
        run<2, 8, true>(src, region, shared, sink);  run<4, 4, true>(src, region, shared, sink);  run<4, 8, true>(src, region, shared, sink);
        run<4, 16, true>(src, region, shared, sink); run<8, 8, true>(src, region, shared, sink);
        run<4, 4, false>(src, region, shared, sink); run<4, 8, false>(src, region, shared, sink); run<8, 8, false>(src, region, shared, sink);
        run<16, 4, false>(src, region, shared, sink);
    }
    printf("-- regions beyond L2 (1 MB per CU = 256 MB in all: Infinity Cache / HBM)\n");
    run<4, 8, true>(src, 1024 * 1024, 0, sink); run<4, 16, true>(src, 1024 * 1024, 0, sink); run<8, 8, false>(src, 1024 * 1024, 0, sink);
    printf("-- 12 KB per CU (a weight k-tile), one region for all CUs\n");
    run<4, 3, true>(src, 12 * 1024, 1, sink); run<4, 3, true>(src, 12 * 1024, 0, sink);
    return 0;
}
