// Batch-wide small ops for gfx950: advantage standardisation, categorical action
// sampling, u8->f32 minibatch gather.
//
//   arl_standardize         <- accel_rl/algos/pg/aac_base.py:136-143
//   arl_sample_categorical  <- rllab/misc/special.py:22-27
//   arl_gather_scale_obs    <- accel_rl/optimizers/util.py:86-89 + policies/layers.py:22-41
//
// All three are HBM-bound streaming kernels (bytes per element in DESIGN.md).

#include <stdarg.h>
#include <string.h>

#include "arl_common.h"

namespace arl {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace arl

extern "C" int arl_abi_version(void) { return ARL_ABI_VERSION; }
extern "C" const char* arl_last_error(void) { return arl::g_err; }

namespace {

constexpr int STD_BLOCKS = 512;     // partial-moment slots

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    return x;
}

// deterministic block reduction of 3 doubles (fixed tree => run-to-run identical)
__device__ __forceinline__ void block_sum3(double& a, double& b, double& c, double* lds) {
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) { lds[w * 3] = a; lds[w * 3 + 1] = b; lds[w * 3 + 2] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sa = 0, sb = 0, sc = 0;
        for (int i = 0; i < nw; ++i) { sa += lds[i * 3]; sb += lds[i * 3 + 1]; sc += lds[i * 3 + 2]; }
        lds[0] = sa; lds[1] = sb; lds[2] = sc;
    }
    __syncthreads();
    a = lds[0]; b = lds[1]; c = lds[2];
    __syncthreads();
}

// pass 1: per-block (sum, sum of squares, count) in f64 -> ws[3*blockIdx]
__global__ __launch_bounds__(256) void moments_kernel(const float* __restrict__ x,
                                                      const int8_t* __restrict__ valids,
                                                      int64_t n, double* __restrict__ ws) {
    __shared__ double lds[16];
    double s = 0, ss = 0, c = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const bool ok = valids ? (valids[i] != 0) : true;
        if (ok) { const double xv = (double)x[i]; s += xv; ss += xv * xv; c += 1.0; }
    }
    block_sum3(s, ss, c, lds);
    if (threadIdx.x == 0) { ws[blockIdx.x * 3] = s; ws[blockIdx.x * 3 + 1] = ss; ws[blockIdx.x * 3 + 2] = c; }
}

// pass 2: every block folds the partials (fixed order), then normalises its share.
__global__ __launch_bounds__(256) void standardize_kernel(float* __restrict__ x,
                                                          const int8_t* __restrict__ valids,
                                                          int64_t n, const double* __restrict__ ws,
                                                          int n_partials, float eps) {
    __shared__ double lds[16];
    double s = 0, ss = 0, c = 0;
    for (int i = threadIdx.x; i < n_partials; i += blockDim.x) { s += ws[i * 3]; ss += ws[i * 3 + 1]; c += ws[i * 3 + 2]; }
    block_sum3(s, ss, c, lds);
    if (c <= 0.0) return;
    const double mean = s / c;
    double var = ss / c - mean * mean;               // population variance (ddof = 0)
    if (var < 0.0) var = 0.0;
    const float mean32 = (float)mean;                // numpy: f32 mean / std of an f32 array
    const float denom = (float)sqrt(var) + eps;      // aac_base.py:139 (std + 1e-6, f32)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const bool ok = valids ? (valids[i] != 0) : true;
        if (ok) x[i] = (x[i] - mean32) / denom;
    }
}

// one lane per row; the cumsum must stay sequential fp32 (special.py:23)
__global__ __launch_bounds__(256) void sample_kernel(const float* __restrict__ prob,
                                                     const double* __restrict__ u, int64_t batch,
                                                     int A, uint8_t* __restrict__ act) {
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < batch;
         b += (int64_t)gridDim.x * blockDim.x) {
        const float* p = prob + b * A;
        const double r = u[b];
        float c = 0.f;
        int k = 0;
        for (int j = 0; j < A; ++j) {
            c += p[j];                               // f32 running sum
            k += ((double)c < r) ? 1 : 0;            // special.py:25 (f32 promoted to f64)
        }
        act[b] = (uint8_t)(k < A - 1 ? k : A - 1);   // special.py:26-27
    }
}

// 16 source bytes per lane -> 4 float4 stores; rows are 16-byte multiples.
__global__ __launch_bounds__(256) void gather_scale_kernel(const uint8_t* __restrict__ obs,
                                                           const int32_t* __restrict__ idx,
                                                           int64_t batch, int64_t row_vec,
                                                           float scale, float* __restrict__ out) {
    const int64_t total = batch * row_vec;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t b = i / row_vec, q = i - b * row_vec;
        const int64_t src_row = idx ? (int64_t)idx[b] : b;
        const uint4 w = reinterpret_cast<const uint4*>(obs)[src_row * row_vec + q];
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
        float4* o = reinterpret_cast<float4*>(out) + i * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o[k] = make_float4((float)(ws[k] & 255u) * scale, (float)((ws[k] >> 8) & 255u) * scale,
                               (float)((ws[k] >> 16) & 255u) * scale, (float)(ws[k] >> 24) * scale);
        }
    }
}

}  // namespace

extern "C" int64_t arl_standardize_workspace_bytes(void) { return (int64_t)STD_BLOCKS * 3 * sizeof(double); }

extern "C" int arl_standardize(float* advantages, const int8_t* valids_or_null, int64_t n,
                               double eps, void* workspace, void* stream) {
    ARL_REQUIRE(advantages && workspace, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(n >= 0, ARL_E_ARG, "negative n");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    int64_t nb = (n + 255) / 256;
    if (nb > STD_BLOCKS) nb = STD_BLOCKS;
    hipLaunchKernelGGL(moments_kernel, dim3((unsigned)nb), dim3(256), 0, s, advantages,
                       valids_or_null, n, (double*)workspace);
    int rc = arl::check_launch("moments_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(standardize_kernel, dim3(arl::stream_grid(n, 256)), dim3(256), 0, s,
                       advantages, valids_or_null, n, (const double*)workspace, (int)nb, (float)eps);
    return arl::check_launch("standardize_kernel");
}

extern "C" int arl_sample_categorical(const float* prob, const double* uniforms, int64_t batch,
                                      int32_t n_actions, uint8_t* actions, void* stream) {
    ARL_REQUIRE(prob && uniforms && actions, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(batch >= 0 && n_actions > 0, ARL_E_ARG, "bad batch/n_actions");
    ARL_REQUIRE(n_actions <= 256, ARL_E_RANGE, "n_actions > 256 needs a wider action dtype");
    if (batch == 0) return 0;
    hipLaunchKernelGGL(sample_kernel, dim3(arl::stream_grid(batch, 256)), dim3(256), 0,
                       (hipStream_t)stream, prob, uniforms, batch, (int)n_actions, actions);
    return arl::check_launch("sample_kernel");
}

// Small staging copies as KERNEL nodes: the per-batch host hand-offs (action uniforms, minibatch permutations, the
// learning-rate multiplier in; completed-episode records and gradient norms out) live in pinned host memory, which the
// device addresses directly.  As memcpy nodes inside the hipGraphs each of them was a 4.4 us blit launch with 6-14 us of
// idle in front of it (the queue switches engines; profiles/r04/bench_steady_state_summary.txt: every idle gap of a step
// sits in front of a copy node).
__global__ __launch_bounds__(256) void copy_words_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16,
                                                         const unsigned char* __restrict__ src_tail,
                                                         unsigned char* __restrict__ dst_tail, int tail) {
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = i0; i < n16; i += stride) dst[i] = src[i];
    if (i0 < tail) dst_tail[i0] = src_tail[i0];
}
__global__ __launch_bounds__(256) void copy_bytes_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                         int64_t n) {
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = i0; i < n; i += stride) dst[i] = src[i];
}

// Append `n` floats to slot counter % n_slots of a ring and advance the device-side counter: an update's diagnostics
// leave a captured hipGraph without an eager copy between two graph replays (the host knows the slot: it counts replays).
__global__ __launch_bounds__(256) void ring_append_kernel(const float* __restrict__ src, int n, float* __restrict__ ring,
                                                          int n_slots, int32_t* counter) {
    const int c = counter[0];
    const int slot = ((c % n_slots) + n_slots) % n_slots;      // (a counter handed in out of range still names a slot)
    for (int i = threadIdx.x; i < n; i += 256) ring[(int64_t)slot * n + i] = src[i];
    __syncthreads();
    if (threadIdx.x == 0) counter[0] = (slot + 1) % n_slots;   // kept reduced: never overflows, however long the run
}

extern "C" int arl_ring_append(const float* src, int32_t n, float* ring, int32_t n_slots, int32_t* counter, void* stream) {
    ARL_REQUIRE(src && ring && counter, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(n > 0 && n_slots > 0, ARL_E_RANGE, "non-positive size");
    hipLaunchKernelGGL(ring_append_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, src, (int)n, ring, (int)n_slots, counter);
    return arl::check_launch("ring_append_kernel");
}

extern "C" int arl_copy_bytes(void* dst, const void* src, int64_t nbytes, void* stream) {
    ARL_REQUIRE(nbytes >= 0 && (nbytes == 0 || (dst && src)), ARL_E_ARG, "null pointer / negative size");
    if (nbytes == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = !((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15);
    if (vec) {
        const int64_t n16 = nbytes >> 4;
        const int tail = (int)(nbytes & 15);
        int64_t grid = (n16 + 255) / 256;
        grid = grid < 1 ? 1 : (grid > 1024 ? 1024 : grid);
        hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)grid), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, n16,
                           (const unsigned char*)src + (n16 << 4), (unsigned char*)dst + (n16 << 4), tail);
    } else {
        int64_t grid = (nbytes + 255) / 256;
        grid = grid > 1024 ? 1024 : grid;
        hipLaunchKernelGGL(copy_bytes_kernel, dim3((unsigned)grid), dim3(256), 0, s, (const unsigned char*)src,
                           (unsigned char*)dst, nbytes);
    }
    return arl::check_launch("copy_kernel");
}

extern "C" int arl_gather_scale_obs(const uint8_t* obs, const int32_t* idx_or_null, int64_t batch,
                                    int64_t row_bytes, float scale, float* out, void* stream) {
    ARL_REQUIRE(obs && out, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(batch >= 0 && row_bytes > 0, ARL_E_ARG, "bad batch/row_bytes");
    ARL_REQUIRE((row_bytes & 15) == 0, ARL_E_RANGE, "row_bytes must be a multiple of 16");
    ARL_REQUIRE(arl::aligned16(obs) && arl::aligned16(out), ARL_E_ALIGN, "obs/out must be 16-byte aligned");
    if (batch == 0) return 0;
    const int64_t row_vec = row_bytes >> 4;
    hipLaunchKernelGGL(gather_scale_kernel, dim3(arl::stream_grid(batch * row_vec, 256)), dim3(256),
                       0, (hipStream_t)stream, obs, idx_or_null, batch, row_vec, scale, out);
    return arl::check_launch("gather_scale_kernel");
}
