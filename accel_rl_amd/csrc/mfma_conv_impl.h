// The policy network's dense contractions on the gfx950 matrix cores, fp32 in / fp32
// accumulate (v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered fmaf chain, so results do
// not depend on launch timing; every split reduction below is folded in a fixed order).
//
// The reference expresses these as Theano conv2d / dot nodes and their gradients
// (accel_rl/policies/pg/networks/pg_cnn.py:45-86, policies/layers.py:22-41,
//  optimizers/single/ppo_optimizer.py:38-56); this file is the MI355X-native form:
//
//   arl_conv2d_fwd         y = relu(conv(x, w) + b)          implicit GEMM, rows gathered on the fly
//   arl_conv2d_bwd_data    dx = conv^T(dy, w) [* (act > 0)]  implicit GEMM per stride-parity class
//   arl_conv2d_bwd_weight  dw = sum_m dy[m]^T im2col(x)[m]   split over m, fixed-order fold
//
// A dense layer is the 1x1 convolution on a 1x1 image (H = W = kh = kw = 1, C = fan_in).
// Layouts: activations NHWC fp32, weights (K, kh, kw, C) ("OHWI", correlation kernels),
// gradients in the same layouts.  All channel counts are multiples of 4 so that every
// gathered fragment is one aligned 16-byte load.
//
// Tiling: 256-thread workgroups = 4 waves; a wave owns TM x TN MFMA tiles of 32 x 32.
// Operand tiles are double-buffered in LDS; global loads for tile k+1 are issued before
// the MFMAs of tile k and written to LDS after them (one barrier per k-tile).  LDS tiles
// whose reduction index is contiguous are padded to BK+4 floats per row so that the
// ds_read_b128 fragment reads (4 consecutive k per lane -> 4 MFMAs) are conflict-free.

#pragma once
#include "arl_optim_dev.h"

#ifndef ARL_PIN_ARGS
#define ARL_PIN_ARGS 1      // development switch (A/B builds); see pin_gemm_args
#endif
#ifndef ARL_WGRAD_INTERLEAVE
#define ARL_WGRAD_INTERLEAVE 1   // development switch (A/B builds); see wgrad_fast_body's k_tile
#endif
#ifndef ARL_AHEAD2
#define ARL_AHEAD2 0        // development switch (A/B builds: ARL_HIPCC_FLAGS=-DARL_AHEAD2=1); measured SLOWER, see igemm_body
#endif

#include <stdlib.h>
#include <type_traits>

namespace arlc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// GEMM rows gathered from an NHWC tensor: row m = (b, oy, ox); reduction index
// r = (ty * taps_x + tx) * Cs + ch reads src[b][y0 + step*ty][x0 + step*tx][ch],
// (y0, x0) = (oy*mul + add_y, ox*mul + add_x); out-of-image taps read 0.
struct GatherDesc {
    const float* src;
    unsigned src_bytes;
    int Hs, Ws, Cs;
    int out_h, out_w;
    int mul, add_y, add_x;
    int taps_x, step;
    // fast path only: taps_y; rmin = smallest tap-origin element offset of a valid row,
    // dmin = smallest tap displacement; origin = rmin + dmin (descriptor base shift, <= 0);
    // src_bytes is then the descriptor size measured from src + origin
    int taps_y, rmin, dmin, origin;
    // fast path only: ceil(2^32 / out_w), ceil(2^32 / out_h) when rows * divisor < 2^32 (then
    // __umulhi(n, magic) == n / divisor exactly for every row index n), else 0 = divide
    unsigned mg_w, mg_h;
    // U8 kernels only (conv 1 straight from the sampler's observations, no f32 copy): planar u8 images,
    // element (b, ch, y, x) = src8[row(b) * img_bytes + ch * plane + y * Ws + x], row(b) = idx ? idx[b] : b;
    // reduction index r = (ch * kh8 + ty) * kw8 + tx (the weights are then (K, C, kh, kw));
    // operand value = float(byte) * scale, converted between the global load and the LDS store
    const unsigned char* src8;
    const int* idx;
    float scale;
    int plane, img_bytes, kh8, kw8;
};

// The weight operand.  B_KC:  element (n, r) at w[n*ld + r]                (r contiguous)
//                      !B_KC: element (r, n) at w[(r % kc)*ld + tap(r / kc) + n], with
// tap(t) = ((i0 + si*(t / taps_x))*kw + (j0 + si*(t % taps_x)))*c          (n contiguous)
struct WeightDesc {
    const float* w;
    unsigned w_bytes;
    int ld, kc, taps_x, i0, j0, si, kw, c;
};

struct OutDesc {
    float* out;
    const float* bias;      // [N] or null
    const float* mask;      // same layout as out; out = 0 where mask <= 0 (relu backward), or null
    unsigned out_bytes;     // size of the whole output tensor (strided epilogue's buffer descriptor)
    int relu, dense;        // dense: out[m*N + n]
    int OH, OW, omul, oadd_y, oadd_x;   // else out[((b*OH + oy*omul + oadd_y)*OW + ox*omul + oadd_x)*N + n]
};

struct GemmArgs {
    GatherDesc g;
    WeightDesc b;
    OutDesc o;
    int M, N, K;
    int k_per_split;        // multiple of BK; gridDim.z splits
    int64_t split_stride;   // elements between split outputs (dense M*N)
    unsigned long long* trace;  // tuning aid: per-workgroup timestamps (arl_dev_conv_trace_buffer), or null
    // stride-s data gradient: the s*s input-pixel parity classes are independent implicit GEMMs that
    // differ only in the fields below; one launch runs them all, blockIdx.z = class (igemm_kernel only)
    int n_par;
    int xcd;                // split kernels: tiles dealt to the XCDs in contiguous ranges (xcd_chunk)
    struct Parity {
        int M, out_h, out_w, add_y, add_x, rmin, dmin, origin, i0, j0, oadd_y, oadd_x;
        unsigned src_bytes, mg_w, mg_h;
    } par[4];
};

// Kernel arguments in ONE round trip.  hipcc loads a by-value argument struct lazily, field by field, in whichever basic
// block first needs it, each s_load followed by its own s_waitcnt: the prologue of igemm_split_kernel made ten dependent
// trips to the kernarg segment (~250-950 cycles each inside a hipGraph: tools/proto/kernarg_probe.hip) before it issued its
// first operand load -- 3 000-3 600 of a 5 400-cycle prologue (tools/prologue_stamps.py).  Naming the scalars a prologue
// needs in one empty asm statement at the top makes the compiler fetch them all at once (one batch of s_loads, one wait).
#define ARL_ARG1(x) asm volatile("" :: "s"(x))
__device__ __forceinline__ void pin_gemm_args(const struct GemmArgs& a);

// XCD-aware placement.  The dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs, each with its own
// 4 MB L2: tiles that share an operand panel (the column tiles of one weight-gradient split, the 64 tiles of one
// forward split of a dense layer, the row tiles over one weight panel) and therefore have neighbouring ids end up on
// eight different L2s, and every one of them pulls the panel over the fabric again -- measured 6-7 TB/s of L1 <- L2
// requests, almost all L2 misses, in kernels whose unique operands are 28-45 MB.  Workgroup `id` of `n` takes tile
// xcd_chunk(id, n): XCD x gets a CONTIGUOUS range of tile ids (bijective for any n).
__device__ __forceinline__ int xcd_chunk(int id, int n) {
    const int q = n >> 3, r = n & 7, x = id & 7, j = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

__device__ __forceinline__ void pin_gemm_args(const GemmArgs& a) {
    asm volatile("" :: "s"(a.g.src), "s"(a.g.src_bytes), "s"(a.g.Hs), "s"(a.g.Ws), "s"(a.g.Cs), "s"(a.g.out_h), "s"(a.g.out_w),
                 "s"(a.g.mul), "s"(a.g.add_y), "s"(a.g.add_x), "s"(a.g.taps_x), "s"(a.g.step), "s"(a.g.taps_y), "s"(a.g.rmin),
                 "s"(a.g.dmin), "s"(a.g.origin), "s"(a.g.mg_w), "s"(a.g.mg_h));
    asm volatile("" :: "s"(a.b.w), "s"(a.b.w_bytes), "s"(a.b.ld), "s"(a.b.kc), "s"(a.b.taps_x), "s"(a.b.i0), "s"(a.b.j0),
                 "s"(a.b.si), "s"(a.b.kw), "s"(a.b.c));
    asm volatile("" :: "s"(a.o.out), "s"(a.o.bias), "s"(a.o.mask), "s"(a.o.out_bytes), "s"(a.o.relu), "s"(a.o.dense), "s"(a.o.OH),
                 "s"(a.o.OW), "s"(a.o.omul), "s"(a.o.oadd_y), "s"(a.o.oadd_x), "s"(a.M), "s"(a.N), "s"(a.K), "s"(a.k_per_split),
                 "s"(a.split_stride), "s"(a.trace), "s"(a.n_par), "s"(a.xcd));
}

// Hardware-bounds-checked 16-byte loads: a raw buffer load whose byte offset lies outside
// the descriptor's range returns 0 and touches no memory, so padding taps, ragged rows and
// the tail of the reduction need neither branches nor selects (the k-loop stays one basic
// block and the scheduler can interleave address math and loads with the MFMAs).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x7ffffff0u;       // > any supported tensor size (checked on the host)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// Row-major epilogue of one wave's TM x TN accumulator tiles: D[row][col] with col = lane & 31,
// row = (v & 3) + 8 (v >> 2) + 4 (lane >> 5).  Each store instruction writes two 128-byte row
// segments; the row part of the address is a compile-time multiple of the row pitch and rides
// in the scalar offset, so the epilogue costs no address arithmetic on the vector unit.
template <int TM, int TN>
__device__ __forceinline__ void store_tiles_rowmajor(const f32x16 (&acc)[TM][TN], float* out, int rows_total,
                                                     int N, int row_base, int col_base, int lane,
                                                     const float* bias, int relu) {
    const int l31 = lane & 31, half = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(out, (unsigned)rows_total * (unsigned)N * 4u);
    const int row0 = row_base + 4 * half;
    const bool full = row_base + TM * 32 <= rows_total;                 // uniform per wave
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = col_base + j * 32 + l31;
        const float bj = (bias && n < N) ? bias[n] : 0.f;
        const unsigned voff = n < N ? (unsigned)(row0 * N + n) << 2 : OOB;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int rc = i * 32 + (v & 3) + 8 * (v >> 2);
                float val = acc[i][j][v] + bj;
                if (relu) val = fmaxf(val, 0.f);
                if (full || row0 + rc < rows_total)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), rs, voff, (unsigned)(rc * N) << 2, 0);
            }
    }
}

// bf16 pieces of fp32 numbers (see SPLIT below): x = h + m + l exactly, h = top 16 bits of x, m = top 16 bits of x - h
constexpr unsigned HI16 = 0xffff0000u;
// (bf16 of x0, bf16 of x1) truncated, x0 in the low half (k order = memory order)
__device__ __forceinline__ unsigned hi_pair(float x0, float x1) {
    return __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
}
__device__ __forceinline__ float lo_part(float x) { return x - __uint_as_float(__float_as_uint(x) & HI16); }   // exact
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = hi_pair(x0, x1);
    const float r0 = lo_part(x0), r1 = lo_part(x1);
    m = hi_pair(r0, r1);
    l = hi_pair(lo_part(r0), lo_part(r1));
}
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Epilogue of the operand-swapped kernels (acc = W-tile x X-tile^T): D'[row][col] with col = lane & 31
// the GEMM row m and row = (v & 3) + 8 (v >> 2) + 4 (lane >> 5) the output channel n, so a lane holds
// four consecutive channels of ONE output row per register quad and stores them as one b128 -- four
// store instructions per 32x32 tile instead of sixteen, and one row decode per lane instead of
// sixteen.  row_off[i] = element offset of the lane's row in tile i (or < 0: row out of range);
// bias_q[j][q] = the lane's four bias values of quad q of column tile j (zeros without a bias);
// N % 4 == 0 (checked on the host).  mask: same layout as out, out = 0 where mask <= 0.
template <int TM, int TN, bool SCALED = false>
__device__ __forceinline__ void store_tiles_quads(const f32x16 (&acc)[TM][TN], __amdgpu_buffer_rsrc_t rs,
                                                  const long long (&row_off)[TM], int N, int col_base, int lane,
                                                  const float4 (&bias_q)[TN][4], const float* mask, int relu,
                                                  const float4 (*pre)[TM] = nullptr, float scale = 1.f) {
    const int half = lane >> 5;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        // the rectifier mask of a column tile: every load issued before the first one is consumed (one latency
        // per column tile instead of one per store; per-workgroup timestamps had the epilogue of the stride-2
        // data gradient at 10 k cycles of a 66 k lifetime)
        float4 mk[4][TM];
        if (mask && pre) {                              // (TN == 1: loaded in the prologue, see igemm_body)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i) mk[q][i] = pre[q][i];
        } else if (mask) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int n = col_base + j * 32 + 8 * q + 4 * half;
                    mk[q][i] = make_float4(1.f, 1.f, 1.f, 1.f);
                    if (row_off[i] >= 0 && n < N) mk[q][i] = *reinterpret_cast<const float4*>(mask + row_off[i] + n);
                }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = col_base + j * 32 + 8 * q + 4 * half;
            const float4 bq = bias_q[j][q];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const bool ok = row_off[i] >= 0 && n < N;
                float4 val = SCALED     // (the u8 kernels: the pixel scale on the finished sum, see bytes_to_f4)
                    ? make_float4(acc[i][j][4 * q] * scale + bq.x, acc[i][j][4 * q + 1] * scale + bq.y,
                                  acc[i][j][4 * q + 2] * scale + bq.z, acc[i][j][4 * q + 3] * scale + bq.w)
                    : make_float4(acc[i][j][4 * q] + bq.x, acc[i][j][4 * q + 1] + bq.y,
                                  acc[i][j][4 * q + 2] + bq.z, acc[i][j][4 * q + 3] + bq.w);
                if (relu) { val.x = fmaxf(val.x, 0.f); val.y = fmaxf(val.y, 0.f); val.z = fmaxf(val.z, 0.f); val.w = fmaxf(val.w, 0.f); }
                const unsigned voff = ok ? (unsigned)((row_off[i] + n) << 2) : OOB;
                if (mask) {
                    const float4 m = mk[q][i];
                    if (!(m.x > 0.f)) val.x = 0.f;
                    if (!(m.y > 0.f)) val.y = 0.f;
                    if (!(m.z > 0.f)) val.z = 0.f;
                    if (!(m.w > 0.f)) val.w = 0.f;
                }
                u32x4 raw = {__float_as_uint(val.x), __float_as_uint(val.y), __float_as_uint(val.z), __float_as_uint(val.w)};
                __builtin_amdgcn_raw_buffer_store_b128(raw, rs, voff, 0, 0);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// out[M][N] = rows(gather)[M][K] . W        (forward conv / dense forward: B_KC;
//                                            data gradient / dense dx: !B_KC)
// TAP_UNIFORM (!B_KC only): kc % BK == 0, so one k-tile lies inside one filter tap and the
// weight-row decode is done once per tile instead of once per loaded row.
// ------------------------------------------------------------------------------------------
template <int WGM, int WGN, int TM, int TN, int BK, bool B_KC, bool TAP_UNIFORM>
__global__ __launch_bounds__(256) void rowgather_gemm_kernel(const GemmArgs a) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32, CH = BK / 4;
    constexpr int LDA = BK + 4;
    constexpr int LDB = B_KC ? BK + 4 : BN;
    constexpr int A_SZ = BM * LDA, B_SZ = B_KC ? BN * LDB : BK * LDB;
    constexpr int ROWS_PER_PASS = 256 / CH;
    constexpr int RA = BM / ROWS_PER_PASS;
    constexpr int NB4 = B_KC ? BN * CH : BK * BN / 4;
    constexpr int RB = (NB4 + 255) / 256;
    static_assert(WGM * WGN == 4 && BM % ROWS_PER_PASS == 0 && BK % 8 == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;
    float* sB = smem + 2 * A_SZ;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, half = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int kbeg = blockIdx.z * a.k_per_split;
    const int kend = (kbeg + a.k_per_split < a.K) ? kbeg + a.k_per_split : a.K;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(a.g.src, a.g.src_bytes);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(a.b.w, a.b.w_bytes);

    // ---- loop-invariant decode of this thread's A rows: tap origin (ry, rx) and element offset of it
    const int a_chunk = tid % CH, a_row0 = tid / CH;
    int ry[RA], rx[RA], rbase[RA];
#pragma unroll
    for (int p = 0; p < RA; ++p) {
        const int m = m0 + a_row0 + p * ROWS_PER_PASS;
        const int t = m / a.g.out_w, ox = m - t * a.g.out_w;
        const int b = t / a.g.out_h, oy = t - b * a.g.out_h;
        ry[p] = m < a.M ? oy * a.g.mul + a.g.add_y : -(1 << 28);
        rx[p] = ox * a.g.mul + a.g.add_x;
        rbase[p] = ((b * a.g.Hs + ry[p]) * a.g.Ws + rx[p]) * a.g.Cs;
    }
    float4 va[RA], vb[RB];
    unsigned offA[RA], offB[RB];        // byte offsets of the NEXT tile's loads (OOB = reads as zero)

    auto plan_tiles = [&](int kb) {
        {
            const int r = kb + a_chunk * 4;
            const int tap = r / a.g.Cs, ch = r - tap * a.g.Cs;
            const int ty = tap / a.g.taps_x, tx = tap - ty * a.g.taps_x;
            const int dy = a.g.step * ty, dx = a.g.step * tx;
            const int delta = (dy * a.g.Ws + dx) * a.g.Cs + ch;
            const int kval = r < kend;
#pragma unroll
            for (int p = 0; p < RA; ++p) {
                const int ok = kval & ((unsigned)(ry[p] + dy) < (unsigned)a.g.Hs) & ((unsigned)(rx[p] + dx) < (unsigned)a.g.Ws);
                offA[p] = ok ? (unsigned)(rbase[p] + delta) << 2 : OOB;
            }
        }
        if (B_KC) {
#pragma unroll
            for (int p = 0; p < RB; ++p) {
                const int idx = tid + p * 256;
                const int nl = idx / CH, chunk = idx - nl * CH;
                const int n = n0 + nl, r = kb + chunk * 4;
                const int ok = (NB4 % 256 == 0 || idx < NB4) & (n < a.N) & (r < kend);
                offB[p] = ok ? (unsigned)(n * a.b.ld + r) << 2 : OOB;
            }
        } else {
            constexpr int NC4 = BN / 4;
            int tile_off = 0;
            if (TAP_UNIFORM) {                  // (kb .. kb+BK) shares one tap
                const int t = kb / a.b.kc, ko0 = kb - t * a.b.kc;
                const int ti = t / a.b.taps_x, tj = t - ti * a.b.taps_x;
                tile_off = ko0 * a.b.ld + ((a.b.i0 + a.b.si * ti) * a.b.kw + (a.b.j0 + a.b.si * tj)) * a.b.c;
            }
#pragma unroll
            for (int p = 0; p < RB; ++p) {
                const int idx = tid + p * 256;
                const int kl = idx / NC4, nch = idx - kl * NC4;
                const int r = kb + kl, n = n0 + nch * 4;
                const int ok = (NB4 % 256 == 0 || idx < NB4) & (r < kend) & (n < a.N);
                int row_off;
                if (TAP_UNIFORM) {
                    row_off = tile_off + kl * a.b.ld;
                } else {
                    const int t = r / a.b.kc, ko = r - t * a.b.kc;
                    const int ti = t / a.b.taps_x, tj = t - ti * a.b.taps_x;
                    row_off = ko * a.b.ld + ((a.b.i0 + a.b.si * ti) * a.b.kw + (a.b.j0 + a.b.si * tj)) * a.b.c;
                }
                offB[p] = ok ? (unsigned)(row_off + n) << 2 : OOB;
            }
        }
    };
    auto issue_loads = [&]() {
#pragma unroll
        for (int p = 0; p < RA; ++p) va[p] = buf_ld4(rsA, offA[p]);
#pragma unroll
        for (int p = 0; p < RB; ++p) vb[p] = buf_ld4(rsB, offB[p]);
    };
    auto store_tiles = [&](int buf) {
        float* dA = sA + buf * A_SZ;
        float* dB = sB + buf * B_SZ;
#pragma unroll
        for (int p = 0; p < RA; ++p)
            *reinterpret_cast<float4*>(dA + (a_row0 + p * ROWS_PER_PASS) * LDA + a_chunk * 4) = va[p];
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            const int idx = tid + p * 256;
            if (NB4 % 256 != 0 && idx >= NB4) continue;
            if (B_KC) {
                const int nl = idx / CH, chunk = idx - nl * CH;
                *reinterpret_cast<float4*>(dB + nl * LDB + chunk * 4) = vb[p];
            } else {
                constexpr int NC4 = BN / 4;
                const int kl = idx / NC4, nch = idx - kl * NC4;
                *reinterpret_cast<float4*>(dB + kl * LDB + nch * 4) = vb[p];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    // Software pipeline: the loads of tile kt+1 are issued first thing in iteration kt from
    // offsets computed during iteration kt-1; the address math for tile kt+2 then runs in the
    // shadow of tile kt's MFMAs, and the LDS stores (which wait for the loads) come last.
    const int nk = (kend - kbeg + BK - 1) / BK;
    plan_tiles(kbeg);
    issue_loads();
    plan_tiles(kbeg + BK);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        issue_loads();                          // tile kt+1 (past the end: all offsets out of range -> zeros, no traffic)
        __builtin_amdgcn_sched_barrier(0);
        plan_tiles(kbeg + (kt + 2) * BK);
        const float* cA = sA + buf * A_SZ + (wm * TM * 32 + l31) * LDA + half * 4;
        const float* cB = B_KC ? sB + buf * B_SZ + (wn * TN * 32 + l31) * LDB + half * 4
                               : sB + buf * B_SZ + (half * 4) * LDB + wn * TN * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < BK / 8; ++ks) {
            float fa[TM][4], fb[TN][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float4 t = *reinterpret_cast<const float4*>(cA + i * 32 * LDA + ks * 8);
                fa[i][0] = t.x; fa[i][1] = t.y; fa[i][2] = t.z; fa[i][3] = t.w;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (B_KC) {
                    const float4 t = *reinterpret_cast<const float4*>(cB + j * 32 * LDB + ks * 8);
                    fb[j][0] = t.x; fb[j][1] = t.y; fb[j][2] = t.z; fb[j][3] = t.w;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) fb[j][q] = cB[(ks * 8 + q) * LDB + j * 32];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][q], fb[j][q], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: D[row][col], col = lane & 31, row = (v & 3) + 8 (v >> 2) + 4 (lane >> 5)
    float* out = a.o.out + (int64_t)blockIdx.z * a.split_stride;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int m = m0 + wm * TM * 32 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * half;
            if (m >= a.M) continue;
            int64_t orow;
            if (a.o.dense) {
                orow = (int64_t)m * a.N;
            } else {
                const int t = m / a.g.out_w, ox = m - t * a.g.out_w;
                const int b = t / a.g.out_h, oy = t - b * a.g.out_h;
                orow = ((int64_t)(b * a.o.OH + oy * a.o.omul + a.o.oadd_y) * a.o.OW + ox * a.o.omul + a.o.oadd_x) * a.N;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * TN * 32 + j * 32 + l31;
                if (n >= a.N) continue;
                float val = acc[i][j][v];
                if (a.o.bias) val += a.o.bias[n];
                if (a.o.relu) val = fmaxf(val, 0.f);
                if (a.o.mask && !(a.o.mask[orow + n] > 0.f)) val = 0.f;
                out[orow + n] = val;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// part[z][ko][r] = sum_{m in split z} dy[m][ko] * rows(gather)[m][r]    (weight gradient)
// The reduction runs over the gathered rows, so their (b, oy, ox) decode changes every
// k-tile: BK lanes decode one row each, one tile ahead, into a small LDS table that every
// thread reads (two integer divisions per tile and workgroup instead of per load).
// ------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* dy;        // [Mred][K_out]
    GatherDesc g;
    float* part;            // [splits][K_out][N]
    unsigned dy_bytes;
    int K_out, N, Mred;
    int m_per_split;        // multiple of BK
    int adv_b, adv_y, adv_x;    // fast path: 256 rows = adv_b images + adv_y output rows + adv_x pixels
    float* bias_part;       // fast path: [splits][K_out] column sums of dy (the bias gradient's partials), or null
    unsigned long long* trace;  // tuning aid (arl_conv_trace_buffer): per-workgroup timestamps as in GemmArgs, or null
    int xcd;                    // split kernels: tiles dealt to the XCDs in contiguous ranges (xcd_chunk)
};

template <int WGM, int WGN, int TM, int TN, int BK>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int A_SZ = BK * BM, B_SZ = BK * BN;
    constexpr int NA4 = BK * BM / 4, RA = (NA4 + 255) / 256, MC4 = BM / 4;
    constexpr int NC4 = BN / 4, KROWS = 256 / NC4, RB = BK / KROWS;
    static_assert(WGM * WGN == 4 && 256 % NC4 == 0 && BK % KROWS == 0 && BK % 8 == 0 && BK <= 64, "tile shape");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int4 s_row[3][BK];       // per gathered row: y0, x0, element offset of (b, y0, x0, 0); beyond the split: y0 << 0
    float* sA = smem;
    float* sB = smem + 2 * A_SZ;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, half = lane >> 5;
    const int n0 = blockIdx.x * BN, i0 = blockIdx.y * BM;
    const int mbeg = blockIdx.z * a.m_per_split;
    const int mend = (mbeg + a.m_per_split < a.Mred) ? mbeg + a.m_per_split : a.Mred;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(a.dy, a.dy_bytes);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(a.g.src, a.g.src_bytes);

    // ---- loop-invariant decode of this thread's gather column (4 consecutive r)
    const int b_c4 = tid % NC4, b_k0 = tid / NC4;
    const int r = n0 + b_c4 * 4;
    const int rval = r < a.N;
    const int tap = r / a.g.Cs, ch = r - tap * a.g.Cs;
    const int ty = tap / a.g.taps_x, tx = tap - ty * a.g.taps_x;
    const int cy = a.g.step * ty, cx = a.g.step * tx;
    const int cdelta = (cy * a.g.Ws + cx) * a.g.Cs + ch;
    float4 va[RA], vb[RB];
    unsigned offA[RA], offB[RB];

    auto decode_rows = [&](int kb, int slot) {      // lanes 0..BK-1 of wave 0
        if (tid < BK) {
            const int m = kb + tid;
            const int t = m / a.g.out_w, ox = m - t * a.g.out_w;
            const int b = t / a.g.out_h, oy = t - b * a.g.out_h;
            const int y0 = m < mend ? oy * a.g.mul + a.g.add_y : -(1 << 28), x0 = ox * a.g.mul + a.g.add_x;
            s_row[slot][tid] = make_int4(y0, x0, ((b * a.g.Hs + y0) * a.g.Ws + x0) * a.g.Cs, 0);
        }
    };
    auto plan_tiles = [&](int kb, int slot) {
#pragma unroll
        for (int p = 0; p < RA; ++p) {
            const int idx = tid + p * 256;
            const int kl = idx / MC4, c4 = idx - kl * MC4;
            const int m = kb + kl, ko = i0 + c4 * 4;
            const int ok = (NA4 % 256 == 0 || idx < NA4) & (m < mend) & (ko < a.K_out);
            offA[p] = ok ? (unsigned)(m * a.K_out + ko) << 2 : OOB;
        }
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            const int4 e = s_row[slot][b_k0 + p * KROWS];
            const int ok = rval & ((unsigned)(e.x + cy) < (unsigned)a.g.Hs) & ((unsigned)(e.y + cx) < (unsigned)a.g.Ws);
            offB[p] = ok ? (unsigned)(e.z + cdelta) << 2 : OOB;
        }
    };
    auto issue_loads = [&]() {
#pragma unroll
        for (int p = 0; p < RA; ++p) va[p] = buf_ld4(rsA, offA[p]);
#pragma unroll
        for (int p = 0; p < RB; ++p) vb[p] = buf_ld4(rsB, offB[p]);
    };
    auto store_tiles = [&](int buf) {
        float* dA = sA + buf * A_SZ;
        float* dB = sB + buf * B_SZ;
#pragma unroll
        for (int p = 0; p < RA; ++p) {
            const int idx = tid + p * 256;
            if (NA4 % 256 != 0 && idx >= NA4) continue;
            *reinterpret_cast<float4*>(dA + idx * 4) = va[p];            // [kl][c4*4] row-major, ld = BM
        }
#pragma unroll
        for (int p = 0; p < RB; ++p)
            *reinterpret_cast<float4*>(dB + (b_k0 + p * KROWS) * BN + b_c4 * 4) = vb[p];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    // Pipeline (see rowgather_gemm_kernel): loads of tile kt+1 first, then the offsets of tile
    // kt+2 (from the row table written one iteration earlier) and the row decode of tile kt+3
    // in the shadow of tile kt's MFMAs.  Row-table slot = tile % 3: the slot written in
    // iteration kt (tile kt+3 = kt mod 3) was last read in iteration kt-1, before a barrier.
    const int nk = (mend - mbeg + BK - 1) / BK;
    decode_rows(mbeg, 0);
    decode_rows(mbeg + BK, 1);
    decode_rows(mbeg + 2 * BK, 2);
    __syncthreads();
    plan_tiles(mbeg, 0);
    issue_loads();
    plan_tiles(mbeg + BK, 1);
    store_tiles(0);
    __syncthreads();
    int slot = 2;                                       // (kt + 2) % 3
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        issue_loads();                                  // tile kt+1
        __builtin_amdgcn_sched_barrier(0);
        plan_tiles(mbeg + (kt + 2) * BK, slot);
        slot = slot == 2 ? 0 : slot + 1;                // (kt + 3) % 3: also the next iteration's plan slot
        decode_rows(mbeg + (kt + 3) * BK, slot);
        const float* cA = sA + buf * A_SZ + (half * 4) * BM + wm * TM * 32 + l31;
        const float* cB = sB + buf * B_SZ + (half * 4) * BN + wn * TN * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < BK / 8; ++ks) {
            float fa[TM][4], fb[TN][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i][q] = cA[(ks * 8 + q) * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j][q] = cB[(ks * 8 + q) * BN + j * 32];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][q], fb[j][q], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        store_tiles(buf ^ 1);
        __syncthreads();
    }

    float* out = a.part + (int64_t)blockIdx.z * a.K_out * a.N;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int ko = i0 + wm * TM * 32 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * half;
            if (ko >= a.K_out) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * TN * 32 + j * 32 + l31;
                if (n < a.N) out[(int64_t)ko * a.N + n] = acc[i][j][v];
            }
        }
    }
}

// ==========================================================================================
// Scalar-addressed fast path.
//
// On gfx950 the fp32-input MFMA runs at the fp32 VECTOR rate and shares the SIMD's VALU issue:
// every VALU instruction in the k-loop is time taken from the MFMAs (measured on MI355X,
// tools/mfma_mix.hip: 6 v_add per MFMA drop 144 -> 93 TF/s; ds_read / buffer_load cost nothing).
// The kernels below therefore keep the per-tile addressing entirely on the scalar unit: a k-tile
// never straddles filter taps, so its address is   per-thread constant (voffset)  +  per-tile
// uniform (soffset, SALU);  padding taps are switched off with one v_bfe_i32 + v_and_or per row
// from a per-row bit mask built once in the prologue.  Requirements (else the generic kernels
// above are used): K % BK == 0 and either Cs % BK == 0 (one tap per tile) or BK % Cs == 0 with
// whole taps of one filter row per tile (MULTI_TAP: conv 1, 4 channels x 8 taps = 32).
// ==========================================================================================
__device__ __forceinline__ float4 buf_ld4s(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ u32x4 buf_ld4u(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
}
__device__ __forceinline__ u32x2 buf_ld2s(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0);
}
__device__ __forceinline__ unsigned buf_ld1s(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0);
}
// four packed bytes -> four floats (v_cvt_f32_ubyte0..3).  The pixel scale (1/255) is NOT applied here: a
// convolution is linear in its input, so the u8 kernels accumulate sum(x * w) on the exact integers and multiply the
// finished sum once -- conv(x * s, w) = s * conv(x, w) up to the rounding of one multiply per output instead of one
// per operand element (two v_pk_mul_f32 per loaded dword, a third of the loader's vector work).
__device__ __forceinline__ float4 bytes_to_f4(unsigned v) {
    return make_float4((float)(v & 0xffu), (float)((v >> 8) & 0xffu), (float)((v >> 16) & 0xffu), (float)(v >> 24));
}
// n / d for a uniform runtime divisor: one v_mul_hi instead of the ~40-instruction division sequence
// (vector instructions in these kernels are paid for in MFMA issue slots); magic == 0 -> plain division
__device__ __forceinline__ int div_u(int n, int d, unsigned magic) {
    if (magic) return (int)__umulhi((unsigned)n, magic);
    return d == 1 ? n : n / d;                      // uniform branches
}
__device__ __forceinline__ unsigned mask_off(unsigned imask, int bit, unsigned voff) {
    // imask bit set = tap invalid for this row -> force the offset out of range
    return ((unsigned)__builtin_amdgcn_sbfe(imask, bit, 1) & OOB) | voff;
}
// Inverted tap mask of a gathered row whose tap (ty, tx) reads pixel (ry + step*ty, rx + step*tx): bit
// ty*taps_x + tx is SET when that pixel lies outside the Hs x Ws image (taps_y * taps_x <= 32, step = +-1).
// Closed form -- the valid taps of a row are a contiguous range in x and in y -- instead of a loop over the taps:
// these kernels pay for every vector instruction in matrix-pipe issue slots (fp32 MFMA shares the VALU), and the
// loops cost 28 instructions per tap row / column, per gathered row, in every workgroup's prologue and in the
// weight gradient's row table refresh (conv 2 forward: 224 of ~400 non-MFMA vector instructions per workgroup).
__device__ __forceinline__ unsigned low_bits(int n) { return n >= 32 ? ~0u : (1u << n) - 1u; }
__device__ __forceinline__ unsigned tap_mask(int ry, int rx, int Hs, int Ws, int taps_y, int taps_x, int step) {
    // x + step*t in [0, W)  <=>  t in [p, p + W) with p = -x (step = 1) or x - W + 1 (step = -1)
    const int px = step > 0 ? -rx : rx - Ws + 1, py = step > 0 ? -ry : ry - Hs + 1;
    const int xlo = min(max(px, 0), 31), xhi = min(px + Ws, taps_x), ylo = max(py, 0), yhi = min(py + Hs, taps_y);
    const int xn = max(xhi - xlo, 0), yn = max(yhi - ylo, 0);
    unsigned good = low_bits(xn) << xlo;                // one tap row's pattern ...
    for (int sh = taps_x; sh < 32; sh *= 2) good |= good << sh;         // ... over every tap row (uniform trip count)
    return ~(good & (low_bits(yn * taps_x) << min(ylo * taps_x, 31))); // rows [ylo, yhi) keep it, the others are out
}

// ==========================================================================================
// SPLIT: fp32 contractions on the bf16 matrix pipe (arl_conv_precision).
//
// gfx950 has no fast fp32 matrix path: v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate (157 TF/s) and takes the
// SIMD's vector issue with it, while v_mfma_f32_32x32x16_bf16 sustains 2.1-2.4 PF/s next to 4-6 vector instructions
// per MFMA (tools/mfma_bf16_mix.hip).  An fp32 number is EXACTLY the sum of three bf16 numbers (24 significand bits =
// 3 x 8: h = top 16 bits of x, m = top 16 bits of x - h, l = x - h - m, every step exact), so
//     x * y = sum over the nine (or the six largest) products of their pieces,
// each product exact in the fp32 accumulator (8 x 8 significand bits).  SPLIT = 9: all nine -- every product term of
// the fp32 contraction enters the sum exactly, only the accumulation rounds (as it does in the fp32 MFMA chain);
// SPLIT = 6: the terms below 2^-24 |x y| (m l, l m, l l) are dropped.  The split happens between the global load
// and the LDS store (11 vector instructions per pair of elements, hidden under the MFMAs of the co-resident waves);
// LDS holds three bf16 planes per operand tile.  u8 observations are exact in ONE bf16 plane (255 < 2^8): conv 1
// needs three products, not nine.
// LDS images per plane: k-contiguous operands [row][BK] bf16, 16-byte slots XOR-swizzled by the row so that the
// ds_read_b128 fragment reads (lane = row, 8 consecutive k) are conflict-free without padding; k-major operands
// (the data gradient's weights, both operands of the weight gradient) as [BK / 2][col] dwords of (k even, k odd)
// pairs, a fragment = 4 ds_read_b32 -- the loader threads fetch two adjacent k rows and pack them.
// ==========================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// acc += sum of the piece products, smallest terms first.  PA / PB = planes of the two operands (1: exact in bf16)
// (the accumulators of a wave's TM x TN tiles take turns inside each piece pair: no back-to-back dependent MFMAs)
template <int SPLIT, int PA, int PB, bool SWAP, int TM, int TN>
__device__ __forceinline__ void split_products(const u32x4 (&fa)[TM][3], const u32x4 (&fb)[TN][3], f32x16 (&acc)[TM][TN]) {
#pragma unroll
    for (int s = 4; s >= 0; --s)
#pragma unroll
        for (int pa = 0; pa < PA; ++pa) {
            const int pb = s - pa;
            if (pb < 0 || pb >= PB) continue;
            if (SPLIT == 6 && PA == 3 && PB == 3 && s > 2) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = SWAP ? mfma_bf16(fb[j][pb], fa[i][pa], acc[i][j]) : mfma_bf16(fa[i][pa], fb[j][pb], acc[i][j]);
        }
}
// 16-byte slot swizzle of the k-contiguous LDS image: NS = BK / 8 slots per row
template <int NS> __device__ __forceinline__ int kc_swz(int row) { return (row / (16 / NS)) % NS; }

// N16: layers with <= 16 output columns (spec 0's 16-filter conv 1, the data gradient into 16 channels) use
// v_mfma_f32_16x16x4_f32 -- a 32-wide tile would spend half of every MFMA on columns that do not exist.
// A wave then owns TM groups of 16 rows x 16 columns; lane (l & 15, l >> 4) holds row l & 15 and the four
// channels 4 (l >> 4) .. + 3 of each group, again one b128 store per group.
// U8: the gathered operand is read from planar u8 images (GatherDesc::src8): a 4-wide k chunk is four
// consecutive pixels of one filter row = one aligned dword (stride, width and plane size are multiples of 4,
// no padding); a k-tile covers BK / kw8 whole filter rows of one plane, so the tile's address is again
// per-thread constant + per-tile uniform.
template <int WGM, int WGN, int TM, int TN, int BK, bool B_KC, bool MULTI_TAP, bool HAS_PAD, bool N16 = false,
          bool U8 = false, int SPLIT = 0, bool ADIR = false>
__device__ __forceinline__ void igemm_body(const GemmArgs& a, const int bx, const int by, const int bz, float* smem) {
    constexpr bool SP = SPLIT != 0;                 // bf16-split products (see above): other LDS images, other MFMAs
    // ADIR: the gathered operand never enters LDS.  With WGN == 1 a wave owns its 32-row tiles outright, and a lane's
    // MFMA fragment -- row l31, eight consecutive k -- is 32 contiguous bytes of that row in memory: two 16-byte loads
    // per 16 k land where the MFMA reads them (fp32, split in registers).  The
    // split kernels are otherwise LDS-bound: three planes written and read back per operand tile is more LDS time
    // than the nine products take on the matrix pipe (128x32 tiles: ~1 400 LDS cycles against 1 152 per k-tile and CU).
    static_assert(!ADIR || (SP && WGN == 1 && !MULTI_TAP), "direct operand: split kernels, one wave per row tile");
    constexpr int MT = N16 ? 16 : 32;               // rows per MFMA tile
    constexpr int BM = WGM * TM * MT, BN = N16 ? WGN * 16 : WGN * TN * 32, CH = BK / 4;
    constexpr int LDA = BK + 4;
    constexpr int LDB = B_KC ? BK + 4 : (N16 ? BN + 4 : BN);    // + 4: the four k-quads of a 16-wide read hit distinct banks
    static_assert(!N16 || (TN == 1 && BK % 16 == 0), "16-wide tiles: one column tile per wave");
    static_assert(!SP || (!N16 && BK % 16 == 0), "split products: 32-wide tiles, two LDS stages");
    constexpr int ROWS_PER_PASS = 256 / CH;
    constexpr int RA = (BM + ROWS_PER_PASS - 1) / ROWS_PER_PASS;        // BM need not be a multiple of a loader pass:
    constexpr int A_SZ = BM * LDA, B_SZ = B_KC ? BN * LDB : BK * LDB;   // the last pass's surplus rows load and store nothing
    constexpr int NST = 2;                                              // LDS stages
    constexpr int NB4 = B_KC ? BN * CH : BK * BN / 4;
    // split, k-major weights: a loader task = two adjacent k rows of four columns (packed into (k, k + 1) dwords)
    constexpr int NPAIR = (BK / 2) * (BN / 4);
    constexpr int RB = (SP && !B_KC) ? 2 * ((NPAIR + 255) / 256) : (NB4 + 255) / 256;
    static_assert(WGM * WGN == 4 && BK % 8 == 0, "tile shape");
    float* sA = smem;
    float* sB = smem + NST * A_SZ;
    // split images (bytes): PA planes of BM x BK bf16 + 3 planes of BK x BN bf16 per stage
    constexpr int PA = U8 ? 1 : 3, ROWB = BK * 2, NS = BK / 8;
    constexpr int LPA = ADIR ? 0 : PA;              // planes of the gathered operand that live in LDS
    constexpr int SPA = BM * ROWB, SPB = BN * ROWB, STAGE = LPA * SPA + 3 * SPB;
    char* const sS = reinterpret_cast<char*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, half = lane >> 5;
    const int m0 = bx * BM, n0 = by * BN;
    GatherDesc g = a.g;
    int M = a.M, w_i0 = a.b.i0, w_j0 = a.b.j0, oadd_y = a.o.oadd_y, oadd_x = a.o.oadd_x;
    if (a.n_par) {                                  // uniform: this workgroup's parity class
        const GemmArgs::Parity& q = a.par[bz];
        M = q.M; g.out_h = q.out_h; g.out_w = q.out_w; g.add_y = q.add_y; g.add_x = q.add_x;
        g.mg_w = q.mg_w; g.mg_h = q.mg_h;
        g.rmin = q.rmin; g.dmin = q.dmin; g.origin = q.origin; g.src_bytes = q.src_bytes;
        w_i0 = q.i0; w_j0 = q.j0; oadd_y = q.oadd_y; oadd_x = q.oadd_x;
        if (m0 >= M) return;
    }
    const int kbeg = a.n_par ? 0 : bz * a.k_per_split;
    const int kend = (kbeg + a.k_per_split < a.K) ? kbeg + a.k_per_split : a.K;
    const int Cs = g.Cs, taps_x = g.taps_x, Ws = g.Ws, step = g.step;
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0, rt0 = 0;
#ifdef ARL_PROLOGUE_STAMPS
    unsigned long long st_a = 0, st_b = 0;
#endif
    if (a.trace) { tr0 = __builtin_readcyclecounter(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    // descriptor origins: the smallest element offset a valid (row, tap) pair can produce
    const __amdgpu_buffer_rsrc_t rsA = U8 ? make_rsrc(reinterpret_cast<const float*>(g.src8), g.src_bytes)
                                          : make_rsrc(g.src + g.origin, g.src_bytes);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(a.b.w, a.b.w_bytes);

    // ---- per-thread constants -------------------------------------------------------------
    const int a_chunk = tid % CH, a_row0 = tid / CH;
    const int cpr8 = g.kw8 >> 2, rpt8 = U8 ? CH / cpr8 : 0;       // U8: chunks per filter row, filter rows per k-tile
    int tpt = 0, chl = a_chunk * 4;                 // tap within the tile / channel within the tap
    if (MULTI_TAP) { tpt = chl / Cs; chl -= tpt * Cs; }
    unsigned voffA[RA], imask[RA], voffB[RB];
    unsigned voffD[TM], imaskD[TM];                 // ADIR: the lane's own row of each of its wave's row tiles
    unsigned voffD8[TM][BK / 16][2];                // ... U8: the two 4-pixel chunks of the lane's k octet of every 16-k step
    if constexpr (ADIR) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + (wm * TM + i) * 32 + l31;
            const int t = div_u(m, g.out_w, g.mg_w), ox = m - t * g.out_w;
            const int b = div_u(t, g.out_h, g.mg_h), oy = t - b * g.out_h;
            const int ry = oy * g.mul + g.add_y, rx = ox * g.mul + g.add_x;
            const int rbase = ((b * g.Hs + ry) * Ws + rx) * Cs;
            voffD[i] = m < M ? (unsigned)(rbase - g.rmin + half * 8) << 2 : OOB;        // k octet `half` of each 16 k
            imaskD[i] = HAS_PAD ? tap_mask(ry, rx, g.Hs, Ws, g.taps_y, taps_x, step) : 0;
            if constexpr (U8) {                     // chunk c of the k-tile = filter row c / cpr8 (of the tile), pixels 4 (c % cpr8) ..
                const int row = (m < M && g.idx) ? g.idx[b] : b;
#pragma unroll
                for (int ks = 0; ks < BK / 16; ++ks)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int c = 2 * (2 * ks + half) + j, tyl = c / cpr8, txq = c - tyl * cpr8;
                        voffD8[i][ks][j] = m < M ? (unsigned)(row * g.img_bytes + (ry + tyl) * Ws + rx + 4 * txq) : OOB;
                    }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < (ADIR ? 0 : RA); ++p) {
        const int m = m0 + a_row0 + p * ROWS_PER_PASS;
        const bool row_ok = m < M && (BM % ROWS_PER_PASS == 0 || a_row0 + p * ROWS_PER_PASS < BM);
        const int t = div_u(m, g.out_w, g.mg_w), ox = m - t * g.out_w;
        const int b = div_u(t, g.out_h, g.mg_h), oy = t - b * g.out_h;
        const int ry = oy * g.mul + g.add_y, rx = ox * g.mul + g.add_x;
        const int rbase = ((b * g.Hs + ry) * Ws + rx) * Cs;
        voffA[p] = row_ok ? (unsigned)(rbase - g.rmin + tpt * Cs + chl) << 2 : OOB;
        if constexpr (U8) {
            const int tyl = a_chunk / cpr8, txq = a_chunk - tyl * cpr8;
            voffA[p] = OOB;
            if (row_ok) {
                const int row = g.idx ? g.idx[b] : b;
                voffA[p] = (unsigned)(row * g.img_bytes + (ry + tyl) * Ws + rx + 4 * txq);
            }
        }
        imask[p] = 0;
        // bit (ty*taps_x + tx) set <=> tap (ty, tx + tpt) is outside the image
        if (HAS_PAD) imask[p] = tap_mask(ry, rx + step * tpt, g.Hs, Ws, g.taps_y, taps_x, step);
    }
#pragma unroll
    for (int p = 0; p < RB; ++p) {
        const int idx = tid + p * 256;
        if (B_KC) {
            const int nl = idx / CH, chunk = idx - nl * CH;
            const int n = n0 + nl;
            voffB[p] = ((NB4 % 256 == 0 || idx < NB4) && n < a.N) ? (unsigned)(n * a.b.ld + chunk * 4) << 2 : OOB;
        } else if (SP) {                            // passes 2q, 2q + 1: rows 2 kl2, 2 kl2 + 1 of task tid + 256 q
            constexpr int NC4 = BN / 4;
            const int t = tid + (p >> 1) * 256;
            const int kl2 = t / NC4, nch = t - kl2 * NC4;
            const int n = n0 + nch * 4;
            voffB[p] = (t < NPAIR && n < a.N) ? (unsigned)((2 * kl2 + (p & 1)) * a.b.ld + n) << 2 : OOB;
        } else {
            constexpr int NC4 = BN / 4;
            const int kl = idx / NC4, nch = idx - kl * NC4;
            const int n = n0 + nch * 4;
            voffB[p] = ((NB4 % 256 == 0 || idx < NB4) && n < a.N) ? (unsigned)(kl * a.b.ld + n) << 2 : OOB;
        }
    }
    // split: byte offset (inside a plane) of the 8 bytes this thread's 4-k chunk of a k-contiguous row lands on
    auto kc_write_off = [&](int row, int chunk) { return row * ROWB + (((chunk >> 1) ^ kc_swz<NS>(row)) << 4) + ((chunk & 1) << 3); };

    // ---- uniform per-tile state (scalar unit) ------------------------------------------------
    int ty, tx, ch0;
    {
        const int tap = kbeg / Cs;
        ch0 = kbeg - tap * Cs;
        ty = tap / taps_x;
        tx = tap - ty * taps_x;
    }
    if constexpr (U8) {                             // (plane ch0, first filter row ty of the tile); tx unused
        const int khw = g.kh8 * g.kw8;
        ch0 = kbeg / khw;
        ty = (kbeg - ch0 * khw) / g.kw8;
        tx = 0;
    }
    // split products: TWO register sets -- tile j rests in set j & 1 for a whole k-tile before it is split into LDS
    // stage j & 1 under the MFMAs of tile j - 1 (the split is ~130 vector instructions per thread and k-tile: it has to
    // run in the MFMAs' shadow, so its operands must have arrived long before)
    constexpr int NR = SP ? 2 : 1;
    float4 va_[NR][RA], vb_[NR][RB];
    unsigned va8_[NR][RA];
    // ADIR: the gathered operand runs ONE tile ahead of the MFMAs (tap state tyA / txA / ch0A), the weights two (through
    // LDS, as above).  fd_[s]: the pieces of tile j, j = s (mod 2) counted so that the last tile is set 1, in fragment
    // layout [row tile][16-k step][piece]; rd_: the fp32 tile in flight, split into fd_ at the end of the tile before.
    constexpr int DST = BK / 16;
    int tyA = ty, txA = tx, ch0A = ch0;
    // AHEAD2 (round 5; the 64-column layers' kernels, TN >= 2): the direct operand runs TWO tiles ahead -- tile kt + 2 is
    // loaded during tile kt into the register set tile kt left, and the split of tile kt + 1 (loaded a whole tile
    // earlier: landed) can be dealt out between the MFMAs from the first one on.  One tile ahead, the split waits for
    // loads issued at the top of the SAME tile: hipcc then clusters the first ~15 MFMAs of a tile without vector work
    // and the last ~20 with seven vector instructions each (profiles/r05/ring_conv_evidence.md).  + 16 registers: only
    // where the kernel is not at a register cliff (the 128 x 32 kernels lost a resident wave to it in round 2).
    // MEASURED (profiles/r05/ahead2_ab.txt, same box, two rounds): conv 2 / conv 3 forward 34.0 / 35.9 -> 36.9 / 39.1 us,
    // conv 3 data gradient 37.3 -> 38.8, bench line 343.5 -> 332.7 k env-steps/s.  OFF; kept as a switch for the record.
    constexpr bool AHEAD2 = ARL_AHEAD2 && ADIR && !U8 && TM * TN >= 2;
    constexpr int NRD = AHEAD2 ? 2 : 1;
    float4 rd_[NRD][TM][DST][2];
    unsigned rd8_[TM][DST][2];
    u32x4 fd_[2][TM][DST][3];
    auto issue_A = [&](auto rs_c) {
        constexpr int rs = decltype(rs_c)::value;
        if constexpr (U8) {
            const unsigned soff8 = (unsigned)(ch0A * g.plane + tyA * Ws);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int ks = 0; ks < DST; ++ks) {
                    rd8_[i][ks][0] = buf_ld1s(rsA, voffD8[i][ks][0], soff8);
                    rd8_[i][ks][1] = buf_ld1s(rsA, voffD8[i][ks][1], soff8);
                }
            return;
        }
        const unsigned soffA = (unsigned)(step * (tyA * Ws + txA) * Cs + ch0A - g.dmin) << 2;
        const int bit = tyA * taps_x + txA;
        constexpr int rr = AHEAD2 ? rs : 0;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const unsigned off = HAS_PAD ? mask_off(imaskD[i], bit, voffD[i]) : voffD[i];
#pragma unroll
            for (int ks = 0; ks < DST; ++ks) {
                rd_[rr][i][ks][0] = buf_ld4s(rsA, off + ks * 64, soffA);
                rd_[rr][i][ks][1] = buf_ld4s(rsA, off + ks * 64 + 16, soffA);
            }
        }
    };
    auto next_tile_A = [&]() {
        if constexpr (U8) {
            tyA += rpt8;
            if (tyA >= g.kh8) { tyA = 0; ++ch0A; }
            return;
        }
        ch0A += BK;
        if (ch0A >= Cs) {
            ch0A = 0;
            if (++txA >= taps_x) { txA = 0; ++tyA; }
        }
    };
    auto split_A = [&](auto rs_c) {                 // rd_ -> fd_[rs]
        constexpr int rs = decltype(rs_c)::value;
        if constexpr (U8) {                         // 0 .. 255 is exact in bf16: one piece
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int ks = 0; ks < DST; ++ks) {
                    const float4 f0 = bytes_to_f4(rd8_[i][ks][0]), f1 = bytes_to_f4(rd8_[i][ks][1]);
                    fd_[rs][i][ks][0] = u32x4{hi_pair(f0.x, f0.y), hi_pair(f0.z, f0.w), hi_pair(f1.x, f1.y), hi_pair(f1.z, f1.w)};
                }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int ks = 0; ks < DST; ++ks) {
                    const float4 q0 = rd_[AHEAD2 ? rs : 0][i][ks][0], q1 = rd_[AHEAD2 ? rs : 0][i][ks][1];
                    unsigned h[4], m[4], l[4];
                    split_pair(q0.x, q0.y, h[0], m[0], l[0]);
                    split_pair(q0.z, q0.w, h[1], m[1], l[1]);
                    split_pair(q1.x, q1.y, h[2], m[2], l[2]);
                    split_pair(q1.z, q1.w, h[3], m[3], l[3]);
                    fd_[rs][i][ks][0] = u32x4{h[0], h[1], h[2], h[3]};
                    fd_[rs][i][ks][1] = u32x4{m[0], m[1], m[2], m[3]};
                    fd_[rs][i][ks][2] = u32x4{l[0], l[1], l[2], l[3]};
                }
        }
    };
    auto issue_loads = [&](int kk, int rs = 0) {    // tile starting at reduction index kk, tap state (ty, tx, ch0)
        auto& va = va_[rs]; auto& vb = vb_[rs]; auto& va8 = va8_[rs];
        if constexpr (U8) {
            const unsigned soffA = (unsigned)(ch0 * g.plane + ty * Ws);
#pragma unroll
            for (int p = 0; p < (ADIR ? 0 : RA); ++p) va8[p] = buf_ld1s(rsA, voffA[p], soffA);
#pragma unroll
            for (int p = 0; p < RB; ++p) vb[p] = buf_ld4s(rsB, voffB[p], (unsigned)kk << 2);
            return;
        }
        const unsigned soffA = (unsigned)(step * (ty * Ws + tx) * Cs + ch0 - g.dmin) << 2;
        unsigned soffB;
        if (B_KC) soffB = (unsigned)kk << 2;
        else soffB = (unsigned)(ch0 * a.b.ld + ((w_i0 + a.b.si * ty) * a.b.kw + (w_j0 + a.b.si * tx)) * a.b.c) << 2;
        const int bit = ty * taps_x + tx;
#pragma unroll
        for (int p = 0; p < (ADIR ? 0 : RA); ++p) {
            const unsigned off = HAS_PAD ? mask_off(imask[p], bit, voffA[p]) : voffA[p];
            va[p] = buf_ld4s(rsA, off, soffA);
        }
#pragma unroll
        for (int p = 0; p < RB; ++p) vb[p] = buf_ld4s(rsB, voffB[p], soffB);
    };
    auto next_tile = [&]() {
        if constexpr (U8) {
            ty += rpt8;
            if (ty >= g.kh8) { ty = 0; ++ch0; }
            return;
        }
        if (MULTI_TAP) {
            tx += BK / Cs;
            if (tx >= taps_x) { tx = 0; ++ty; }
        } else {
            ch0 += BK;
            if (ch0 >= Cs) {
                ch0 = 0;
                if (++tx >= taps_x) { tx = 0; ++ty; }
            }
        }
    };
    auto store_tiles = [&](int buf, int rs = 0) {
        auto& va = va_[rs]; auto& vb = vb_[rs]; auto& va8 = va8_[rs];
        if constexpr (SP) {
            char* dS = sS + buf * STAGE;
#pragma unroll
            for (int p = 0; p < (ADIR ? 0 : RA); ++p) {
                if (BM % ROWS_PER_PASS != 0 && a_row0 + p * ROWS_PER_PASS >= BM) continue;
                char* d = dS + kc_write_off(a_row0 + p * ROWS_PER_PASS, a_chunk);
                if constexpr (U8) {                        // 0 .. 255 is exact in bf16: one plane
                    const float4 f = bytes_to_f4(va8[p]);
                    *reinterpret_cast<uint2*>(d) = make_uint2(hi_pair(f.x, f.y), hi_pair(f.z, f.w));
                } else {
                    uint2 h, m, l;
                    split_pair(va[p].x, va[p].y, h.x, m.x, l.x);
                    split_pair(va[p].z, va[p].w, h.y, m.y, l.y);
                    *reinterpret_cast<uint2*>(d) = h;
                    *reinterpret_cast<uint2*>(d + SPA) = m;
                    *reinterpret_cast<uint2*>(d + 2 * SPA) = l;
                }
            }
            char* dB = dS + LPA * SPA;
            if constexpr (B_KC) {
#pragma unroll
                for (int p = 0; p < RB; ++p) {
                    const int idx = tid + p * 256;
                    if (NB4 % 256 != 0 && idx >= NB4) continue;
                    const int nl = idx / CH, chunk = idx - nl * CH;
                    char* d = dB + kc_write_off(nl, chunk);
                    uint2 h, m, l;
                    split_pair(vb[p].x, vb[p].y, h.x, m.x, l.x);
                    split_pair(vb[p].z, vb[p].w, h.y, m.y, l.y);
                    *reinterpret_cast<uint2*>(d) = h;
                    *reinterpret_cast<uint2*>(d + SPB) = m;
                    *reinterpret_cast<uint2*>(d + 2 * SPB) = l;
                }
            } else {
#pragma unroll
                for (int q = 0; q < RB / 2; ++q) {
                    constexpr int NC4 = BN / 4;
                    const int t = tid + q * 256;
                    if (NPAIR % 256 != 0 && t >= NPAIR) continue;
                    const int kl2 = t / NC4, nch = t - kl2 * NC4;
                    char* d = dB + (kl2 * BN + nch * 4) * 4;
                    const float4 v0 = vb[2 * q], v1 = vb[2 * q + 1];
                    uint4 h, m, l;
                    split_pair(v0.x, v1.x, h.x, m.x, l.x);
                    split_pair(v0.y, v1.y, h.y, m.y, l.y);
                    split_pair(v0.z, v1.z, h.z, m.z, l.z);
                    split_pair(v0.w, v1.w, h.w, m.w, l.w);
                    *reinterpret_cast<uint4*>(d) = h;
                    *reinterpret_cast<uint4*>(d + SPB) = m;
                    *reinterpret_cast<uint4*>(d + 2 * SPB) = l;
                }
            }
            return;
        }
        float* dA = sA + buf * A_SZ;
        float* dB = sB + buf * B_SZ;
#pragma unroll
        for (int p = 0; p < RA; ++p) {
            if (BM % ROWS_PER_PASS != 0 && a_row0 + p * ROWS_PER_PASS >= BM) continue;
            *reinterpret_cast<float4*>(dA + (a_row0 + p * ROWS_PER_PASS) * LDA + a_chunk * 4) =
                U8 ? bytes_to_f4(va8[p]) : va[p];
        }
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            const int idx = tid + p * 256;
            if (NB4 % 256 != 0 && idx >= NB4) continue;
            if (B_KC) {
                const int nl = idx / CH, chunk = idx - nl * CH;
                *reinterpret_cast<float4*>(dB + nl * LDB + chunk * 4) = vb[p];
            } else {
                constexpr int NC4 = BN / 4;
                const int kl = idx / NC4, nch = idx - kl * NC4;
                *reinterpret_cast<float4*>(dB + kl * LDB + nch * 4) = vb[p];
            }
        }
    };

    const int l15 = lane & 15, quad = lane >> 4;    // N16 lane coordinates
    f32x16 acc[TM][TN];
    f32x4 acc16[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) acc16[i][v] = 0.f;
    }

    // The epilogue's bias: loaded here, consumed after the loop (no loop-carried copies, latency long gone).
    float4 bias_q[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = N16 ? n0 + wn * 16 + 4 * quad : n0 + wn * TN * 32 + j * 32 + 8 * q + 4 * half;
            bias_q[j][q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.o.bias && n < a.N && (!N16 || q == 0)) bias_q[j][q] = *reinterpret_cast<const float4*>(a.o.bias + n);
        }
    // The epilogue's output rows -- and, for the one-tile-per-wave shapes, the rectifier mask of the layer below
    // (a data gradient's epilogue otherwise starts with a dependent global load per store: 8-10 k cycles of a 65 k
    // workgroup lifetime in the stride-2 data gradient) -- are fetched here, a whole main loop ahead of their use.
    long long row_off[TM];
    auto decode_out_rows = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + (wm * TM + i) * MT + (N16 ? l15 : l31);
            if (a.o.dense) {
                row_off[i] = m < M ? (long long)m * a.N : -1;
            } else {                                    // stride-parity data gradient: rows map to scattered pixels
                const int t = div_u(m, g.out_w, g.mg_w), ox = m - t * g.out_w;
                const int b = div_u(t, g.out_h, g.mg_h), oy = t - b * g.out_h;
                row_off[i] = m < M ? ((long long)(b * a.o.OH + oy * a.o.omul + oadd_y) * a.o.OW + ox * a.o.omul + oadd_x) * a.N
                                   : -1;
            }
        }
    };
    constexpr bool PRE_MASK = TN == 1 && TM <= 2;       // (larger register tiles keep their registers for the main loop)
    if (PRE_MASK) decode_out_rows();
    float4 mk_pre[4][TM];
    // issued at the start of the LAST k-tile: behind every operand load (an earlier issue would sit in front of the
    // tile loads in the in-order vmcnt queue and stall the first LDS store on scattered, cache-cold addresses)
    auto issue_mask_loads = [&]() {
        if (!(PRE_MASK && a.o.mask)) return;
#pragma unroll
        for (int q = 0; q < (N16 ? 1 : 4); ++q)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int n = N16 ? n0 + wn * 16 + 4 * quad : n0 + wn * TN * 32 + 8 * q + 4 * half;
                mk_pre[q][i] = make_float4(1.f, 1.f, 1.f, 1.f);
                if (row_off[i] >= 0 && n < a.N) mk_pre[q][i] = *reinterpret_cast<const float4*>(a.o.mask + row_off[i] + n);
            }
    };
    const int nk = (kend - kbeg) / BK;
    // The MFMAs of sub-steps [LO, HI) of the k-tile in LDS stage BUF (16 k per sub-step with the 16-wide tiles, 8
    // otherwise).  Compile-time stage: every LDS address is then a per-thread constant plus an immediate (with a
    // run-time buffer index the compiler re-derived four base addresses per tile with vector adds, and every vector
    // instruction here is taken from the MFMAs' issue slots).
    constexpr int STEPS = (N16 || SP) ? BK / 16 : BK / 8;
    auto mfma_steps = [&](auto buf_c, auto lo_c, auto hi_c) {
        constexpr int buf = decltype(buf_c)::value, LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
        if constexpr (SP) {
            // a lane's fragment = row l31 (of its 32-row tile), k octet 2 ks + half: one 16-byte slot per plane
            const char* cA = sS + buf * STAGE + (wm * TM * 32) * ROWB + l31 * ROWB;
            const char* cB = sS + buf * STAGE + LPA * SPA + (B_KC ? (wn * TN * 32) * ROWB + l31 * ROWB
                                                                  : (half * 4 * BN + wn * TN * 32 + l31) * 4);
            const int swz = kc_swz<NS>(l31);
#pragma unroll
            for (int ks = LO; ks < HI; ++ks) {
                const int slot = ((2 * ks + half) ^ swz) << 4;
                u32x4 fa[TM][3], fb[TN][3];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int pl = 0; pl < PA; ++pl) {
                        if constexpr (ADIR) fa[i][pl] = fd_[buf][i][ks][pl];
                        else fa[i][pl] = *reinterpret_cast<const u32x4*>(cA + pl * SPA + i * 32 * ROWB + slot);
                    }
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        if constexpr (B_KC) {
                            fb[j][pl] = *reinterpret_cast<const u32x4*>(cB + pl * SPB + j * 32 * ROWB + slot);
                        } else {
                            const unsigned* q = reinterpret_cast<const unsigned*>(cB + pl * SPB + (ks * 8 * BN + j * 32) * 4);
                            fb[j][pl] = u32x4{q[0], q[BN], q[2 * BN], q[3 * BN]};
                        }
                    }
                split_products<SPLIT, PA, 3, true, TM, TN>(fa, fb, acc);
            }
        } else if constexpr (N16) {
            const float* cA = sA + buf * A_SZ + (wm * TM * 16 + l15) * LDA + quad * 4;
            const float* cB = B_KC ? sB + buf * B_SZ + (wn * 16 + l15) * LDB + quad * 4
                                   : sB + buf * B_SZ + (quad * 4) * LDB + wn * 16 + l15;
#pragma unroll
            for (int ks = LO; ks < HI; ++ks) {
                float fa[TM][4], fb[4];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float4 t = *reinterpret_cast<const float4*>(cA + i * 16 * LDA + ks * 16);
                    fa[i][0] = t.x; fa[i][1] = t.y; fa[i][2] = t.z; fa[i][3] = t.w;
                }
                if (B_KC) {
                    const float4 t = *reinterpret_cast<const float4*>(cB + ks * 16);
                    fb[0] = t.x; fb[1] = t.y; fb[2] = t.z; fb[3] = t.w;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) fb[q] = cB[(ks * 16 + q) * LDB];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc16[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[q], fa[i][q], acc16[i], 0, 0, 0);
            }
        } else {
            const float* cA = sA + buf * A_SZ + (wm * TM * 32 + l31) * LDA + half * 4;
            const float* cB = B_KC ? sB + buf * B_SZ + (wn * TN * 32 + l31) * LDB + half * 4
                                   : sB + buf * B_SZ + (half * 4) * LDB + wn * TN * 32 + l31;
#pragma unroll
            for (int ks = LO; ks < HI; ++ks) {
                float fa[TM][4], fb[TN][4];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float4 t = *reinterpret_cast<const float4*>(cA + i * 32 * LDA + ks * 8);
                    fa[i][0] = t.x; fa[i][1] = t.y; fa[i][2] = t.z; fa[i][3] = t.w;
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (B_KC) {
                        const float4 t = *reinterpret_cast<const float4*>(cB + j * 32 * LDB + ks * 8);
                        fb[j][0] = t.x; fb[j][1] = t.y; fb[j][2] = t.z; fb[j][3] = t.w;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) fb[j][q] = cB[(ks * 8 + q) * LDB + j * 32];
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j][q], fa[i][q], acc[i][j], 0, 0, 0);
            }
        }
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    using CH_ = std::integral_constant<int, STEPS / 2>;
    using CS_ = std::integral_constant<int, STEPS>;
    if constexpr (SP) {
        // tile j: register set and LDS stage (j + nk) & 1 (the loop ends on stage 1)
#ifdef ARL_PROLOGUE_STAMPS      // development: where a workgroup's prologue goes (tools/prologue_stamps.py, t[6] / t[7])
#define ARL_STAMP(x) do { if (a.trace) { __builtin_amdgcn_sched_barrier(0); x = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define ARL_STAMP(x) do { } while (0)
#endif
        if (nk & 1) {                                   // uniform
            if constexpr (ADIR) issue_A(C1{});
            issue_loads(kbeg, 1);
            if (nk > 1) { next_tile(); issue_loads(kbeg + BK, 0); }
            if constexpr (AHEAD2) { if (nk > 1) { next_tile_A(); issue_A(C0{}); } }      // tile 1
            ARL_STAMP(st_a);
            store_tiles(1, 1);
            if constexpr (ADIR) split_A(C1{});
            ARL_STAMP(st_b);
        } else {
            if constexpr (ADIR) issue_A(C0{});
            issue_loads(kbeg, 0);
            next_tile(); issue_loads(kbeg + BK, 1);
            if constexpr (AHEAD2) { next_tile_A(); issue_A(C1{}); }                       // tile 1 (nk >= 2)
            ARL_STAMP(st_a);
            store_tiles(0, 0);
            if constexpr (ADIR) split_A(C0{});
            ARL_STAMP(st_b);
        }
        __syncthreads();
        if (a.trace) tr1 = __builtin_readcyclecounter();
        // One basic block per steady-state k-tile: the MFMAs of tile kt (LDS stage buf) and the split + LDS stores of
        // tile kt + 1 (register set and stage buf ^ 1) -- left to itself the scheduler issues the MFMAs in one clump
        // and the ~130 vector instructions of the split after them; the group barriers below deal the vector work and
        // the LDS stores out between the MFMAs, where they cost nothing (tools/mfma_bf16_mix.hip: 4-6 per MFMA are free).
        constexpr int NPROD = PA == 1 ? 3 : SPLIT;
        constexpr int NM = TM * TN * NPROD * STEPS;                                     // MFMAs per k-tile and wave
        constexpr int NV = RA * (U8 ? 6 : 22) + (B_KC ? RB * 22 : (RB / 2) * 44)    // the split's vector instructions
                           + (AHEAD2 ? TM * DST * 44 : 0);                           // (+ the direct operand's, where it can start at once)
        constexpr int NW = RA * PA + (B_KC ? RB * 3 : (RB / 2) * 3);                    // its LDS stores
        constexpr int VPM = (NV + NM - 1) / NM < 6 ? (NV + NM - 1) / NM : 6;
        constexpr int WEV = NM / NW > 0 ? NM / NW : 1;
        auto mid_tile = [&](auto buf_c, int kt) {       // tiles 0 .. nk - 2
            constexpr int buf = decltype(buf_c)::value;
            if (kt + 2 < nk) {                          // uniform: tile kt + 2 -> the set tile kt has left
                next_tile();
                issue_loads(kbeg + (kt + 2) * BK, buf);
            }
            if constexpr (AHEAD2) {                     // tile kt + 2 of the direct operand -> the set tile kt has left
                if (kt + 2 < nk) {
                    next_tile_A();
                    issue_A(buf_c);
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma_steps(buf_c, C0{}, CS_{});
                store_tiles(buf ^ 1, buf ^ 1);
                split_A(std::integral_constant<int, (buf ^ 1)>{});     // tile kt + 1: loaded during tile kt - 1
                __builtin_amdgcn_sched_group_barrier(0x100, STEPS * TN * 3, 0);
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                    if (m % WEV == WEV - 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
                __syncthreads();
                return;
            } else if constexpr (ADIR) {                // tile kt + 1 of the direct operand
                next_tile_A();
                issue_A(std::integral_constant<int, (buf ^ 1)>{});
                __builtin_amdgcn_sched_barrier(0);
                mfma_steps(buf_c, C0{}, CS_{});
                store_tiles(buf ^ 1, buf ^ 1);
                split_A(std::integral_constant<int, (buf ^ 1)>{});
                __syncthreads();
                return;
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_steps(buf_c, C0{}, CS_{});
            store_tiles(buf ^ 1, buf ^ 1);
            // (every fragment read first: the LDS stores of the other stage cannot be proven not to alias them and would
            //  otherwise queue up behind the last read, at the end of the tile)
            __builtin_amdgcn_sched_group_barrier(0x100, STEPS * (TM * PA + TN * 3 * (B_KC ? 1 : 4)), 0);
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                if (m % WEV == WEV - 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
            __syncthreads();
        };
        int kt = 0;
        if (!(nk & 1)) { mid_tile(C0{}, 0); kt = 1; }
        for (; kt + 1 < nk; kt += 2) {
            mid_tile(C1{}, kt);
            mid_tile(C0{}, kt + 1);
        }
        issue_mask_loads();
        __builtin_amdgcn_sched_barrier(0);
        mfma_steps(C1{}, C0{}, CS_{});                  // the last tile
    } else {
    issue_loads(kbeg);
    store_tiles(nk & 1);                            // first tile's buffer chosen so that the loop ends on buffer 1
    __syncthreads();
    if (a.trace) tr1 = __builtin_readcyclecounter();
    // One k-tile with a COMPILE-TIME buffer index (see mfma_steps).
    auto k_tile = [&](auto buf_c, int kt) {
        constexpr int buf = decltype(buf_c)::value;
        if (kt + 1 < nk) {                          // uniform branch
            next_tile();
            issue_loads(kbeg + (kt + 1) * BK);
        } else {
            issue_mask_loads();
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_steps(buf_c, C0{}, CS_{});
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) {
            store_tiles(buf ^ 1);
            __syncthreads();
        }
    };
    {   // an odd tile count peels its FIRST tile (from buffer 1); the rest is whole (buffer 0, buffer 1) pairs
        int kt = 0;
        if (nk & 1) { k_tile(C1{}, 0); kt = 1; }
        for (; kt < nk; kt += 2) {
            k_tile(C0{}, kt);
            k_tile(C1{}, kt + 1);
        }
    }
    }

    if (a.trace) tr2 = __builtin_readcyclecounter();
    float* out = a.o.out + (a.n_par ? 0 : (int64_t)bz * a.split_stride);
    {
        // operands are swapped (acc = W-tile x X-tile^T): every lane owns ONE output row per 32-row tile
        const __amdgpu_buffer_rsrc_t rsO = make_rsrc(out, a.o.out_bytes);
        if (!PRE_MASK) decode_out_rows();
        if constexpr (N16) {
            const int n = n0 + wn * 16 + 4 * quad;
            const float4 bq = bias_q[0][0];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const bool ok = row_off[i] >= 0 && n < a.N;
                float4 val = U8 ? make_float4(acc16[i][0] * g.scale + bq.x, acc16[i][1] * g.scale + bq.y,
                                              acc16[i][2] * g.scale + bq.z, acc16[i][3] * g.scale + bq.w)
                                : make_float4(acc16[i][0] + bq.x, acc16[i][1] + bq.y, acc16[i][2] + bq.z, acc16[i][3] + bq.w);
                if (a.o.relu) { val.x = fmaxf(val.x, 0.f); val.y = fmaxf(val.y, 0.f); val.z = fmaxf(val.z, 0.f); val.w = fmaxf(val.w, 0.f); }
                if (a.o.mask) {
                    float4 mk = make_float4(1.f, 1.f, 1.f, 1.f);
                    if (PRE_MASK) mk = mk_pre[0][i < TM ? i : 0];
                    else if (ok) mk = *reinterpret_cast<const float4*>(a.o.mask + row_off[i] + n);
                    if (!(mk.x > 0.f)) val.x = 0.f;
                    if (!(mk.y > 0.f)) val.y = 0.f;
                    if (!(mk.z > 0.f)) val.z = 0.f;
                    if (!(mk.w > 0.f)) val.w = 0.f;
                }
                u32x4 raw = {__float_as_uint(val.x), __float_as_uint(val.y), __float_as_uint(val.z), __float_as_uint(val.w)};
                __builtin_amdgcn_raw_buffer_store_b128(raw, rsO, ok ? (unsigned)((row_off[i] + n) << 2) : OOB, 0, 0);
            }
        } else {
            store_tiles_quads<TM, TN, U8>(acc, rsO, row_off, a.N, n0 + wn * TN * 32, lane, bias_q, a.o.mask, a.o.relu,
                                          PRE_MASK ? mk_pre : nullptr, g.scale);
        }
    }
    if (a.trace && tid == 0) {
        unsigned long long* t = a.trace + ((size_t)(bz * gridDim.y + by) * gridDim.x + bx) * 8;     // plain launches only
        t[0] = tr0; t[1] = tr1; t[2] = tr2; t[3] = __builtin_readcyclecounter();
        t[4] = rt0; t[5] = __builtin_amdgcn_s_memrealtime();
#ifdef ARL_PROLOGUE_STAMPS
        if constexpr (SP) { t[6] = st_a; t[7] = st_b; }
        else
#endif
        {
        t[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID
        t[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // XCC_ID
        }
    }
}

// (Register allocation: the 128-row x 32-column, 16-deep shape takes 100-106 registers = FOUR workgroups per CU.  Capping
// it at 96 for a fifth (__launch_bounds__(256, 5): 4-9 spilled registers) was measured inside the learner: the fifth
// workgroup is resident, the CU's timeline stays at ~105 k cycles for 8 x 8 192 matrix-pipe cycles of work -- with five
// waves per SIMD in their main loops the pipe is still only ~2/3 busy, so residency is not what holds these two kernels
// (conv 1 forward, stride-2 data gradient) back; tools/context_trace.py prints the timelines.)
template <int WGM, int WGN, int TM, int TN, int BK, bool B_KC, bool MULTI_TAP, bool HAS_PAD, bool N16 = false>
__global__ __launch_bounds__(256) void igemm_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    igemm_body<WGM, WGN, TM, TN, BK, B_KC, MULTI_TAP, HAS_PAD, N16>(a, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// bf16-split products (igemm_body, SPLIT): MINW waves per SIMD; CORUN: an optimiser job rides in the grid's first workgroups
// (arl_corun_job), as in igemm_occ_kernel
template <int WGM, int WGN, int TM, int TN, int BK, bool B_KC, bool MULTI_TAP, bool HAS_PAD, bool U8, int SPLIT, int MINW,
          bool CORUN = false, bool ADIR = false>
__global__ __launch_bounds__(256, MINW) void igemm_split_kernel(const GemmArgs a, const arl::OptSeg c) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
#if ARL_PIN_ARGS
    pin_gemm_args(a);
#endif
    int bx = blockIdx.x;
    if constexpr (CORUN) {
        if (bx < c.co_blocks) {
            __shared__ double lds[8];
            if (blockIdx.y || blockIdx.z) return;
            if (c.method == ARL_OPT_ADAM) arl::opt_update_block<ARL_OPT_ADAM>(c, bx, c.co_blocks, lds);
            else arl::opt_update_block<ARL_OPT_RMSPROP>(c, bx, c.co_blocks, lds);
            return;
        }
        bx -= c.co_blocks;
    }
    int by = blockIdx.y, bz = blockIdx.z;
    if (!CORUN && a.xcd) {                          // uniform
        const int gx = gridDim.x, gy = gridDim.y;
        const int t = xcd_chunk((bz * gy + by) * gx + bx, gx * gy * (int)gridDim.z);
        bx = t % gx;
        const int u = t / gx;
        by = u % gy; bz = u / gy;
    }
    igemm_body<WGM, WGN, TM, TN, BK, B_KC, MULTI_TAP, HAS_PAD, false, U8, SPLIT, ADIR>(a, bx, by, bz, smem);
}

template <int WGM, int WGN, int TM, int TN, int BK, bool B_KC, bool MULTI_TAP, bool HAS_PAD, bool N16, int MINW, bool CORUN = false,
          bool U8 = false>
__global__ __launch_bounds__(256, MINW) void igemm_occ_kernel(const GemmArgs a, const arl::OptSeg c) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if constexpr (CORUN) {
        // The update's workgroups come FIRST in the grid -- one per CU, streaming from the start of the launch, while
        // the tile workgroups fill the other four slots of every CU (appended behind the tiles they ran in the tail
        // and lengthened it: +8.8 us inside the learner instead of +1).
        if ((int)blockIdx.x < c.co_blocks) {
            __shared__ double lds[8];
            if (blockIdx.y || blockIdx.z) return;
            if (c.method == ARL_OPT_ADAM) arl::opt_update_block<ARL_OPT_ADAM>(c, (int)blockIdx.x, c.co_blocks, lds);
            else arl::opt_update_block<ARL_OPT_RMSPROP>(c, (int)blockIdx.x, c.co_blocks, lds);
            return;
        }
        igemm_body<WGM, WGN, TM, TN, BK, B_KC, MULTI_TAP, HAS_PAD, N16, U8>(a, blockIdx.x - c.co_blocks, blockIdx.y, blockIdx.z, smem);
        return;
    }
    igemm_body<WGM, WGN, TM, TN, BK, B_KC, MULTI_TAP, HAS_PAD, N16, U8>(a, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// forward convolution straight from planar u8 observations (see igemm_body, U8)
template <int WGM, int WGN, int TM, int TN, int BK, bool N16>
__global__ __launch_bounds__(256) void igemm_u8_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    igemm_body<WGM, WGN, TM, TN, BK, true, false, false, N16, true>(a, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// Weight gradient, scalar-addressed: dy advances by a uniform stride per tile (soffset); the
// gathered rows change every tile, so their element offsets and padding masks come from an LDS
// table that all 256 threads refresh together, 256 rows (= 256 / BK tiles) at a time, each
// thread walking its own row's (b, oy, ox) incrementally (no divisions in the loop).
// Requirements: Mred % 256 == 0 is NOT needed, but Mred % BK == 0 and m_per_split % BK == 0.
constexpr int WG_ROWS = 256;

// M16: <= 16 output channels (spec 0's conv 1) -> v_mfma_f32_16x16x4_f32, 16 channel rows x 16-column groups
// (a 32-row tile would spend half of every MFMA on channels that do not exist).
// U8: the gathered rows come from planar u8 images (GatherDesc::src8; column r = (ch * kh8 + ty) * kw8 + tx,
// so dw is (K, C, kh, kw)); the row count needs no rounding (the last tile's missing rows read as zeros).
// SPLIT: bf16-split products (see igemm_body): both operands are k-major here, so both LDS images are pair-packed --
// every loader task fetches two adjacent reduction rows of its four columns.
template <int WGM, int WGN, int TM, int TN, int BK, bool HAS_PAD, bool M16 = false, bool U8 = false, int SPLIT = 0>
__device__ __forceinline__ void wgrad_fast_body(const WgradArgs& a, const int bx, const int by, const int bz, float* smem) {
    constexpr bool SP = SPLIT != 0;
    constexpr int BM = M16 ? 16 : WGM * TM * 32, BN = WGN * TN * 32;
    static_assert(!M16 || (WGM == 1 && TM == 1 && BK % 16 == 0), "16-row tiles: one row tile");
    static_assert(!SP || (!M16 && BK % 16 == 0), "split products: 32-row tiles");
    constexpr int A_SZ = BK * BM, B_SZ = BK * BN;
    constexpr int MC4 = BM / 4, NPA = (BK / 2) * MC4;        // split: pair tasks of the dy tile
    constexpr int NA4 = BK * BM / 4, RA = SP ? 2 * ((NPA + 255) / 256) : (NA4 + 255) / 256;
    // W8 (u8 observations, split products): a gather task is a whole 8-pixel filter row (ONE 8-byte load) instead of a
    // 4-pixel chunk.  The launch is bound by L1 ACCESSES (tools/conv1_pmc.sh: 16.7 M per launch at the PPO minibatch,
    // 65 k per CU): a wave's load touches the same ~16-20 lines either way -- one per (plane, filter row) of its columns
    // -- so twice the bytes per instruction halves them.  Needs kw % 8 == 0 (the dispatcher's condition).
    constexpr bool W8 = U8 && SP;
    constexpr int CW = W8 ? 8 : 4;                           // columns per gather task
    constexpr int NC4 = BN / CW, KROWS = 256 / NC4, RB = BK / KROWS;
    static_assert(!SP || RB % 2 == 0, "split products: an even number of gather passes (row pairs)");
    constexpr int PB = U8 ? 1 : 3;
    constexpr int SPA = BK * BM * 2, SPB = BK * BN * 2, STAGE = 3 * SPA + PB * SPB;   // bytes per plane / stage
    char* const sS = reinterpret_cast<char*>(smem);
    constexpr int TILES_PER_GROUP = WG_ROWS / BK;
    static_assert(WGM * WGN == 4 && 256 % NC4 == 0 && BK % KROWS == 0 && BK % 8 == 0 && WG_ROWS % BK == 0, "tile shape");
    __shared__ uint2 s_row[2][WG_ROWS];     // per gathered row: byte offset of its tap origin, inverted tap mask
    float* sA = smem;
    float* sB = smem + 2 * A_SZ;
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0, rt0 = 0;
    if (a.trace) { tr0 = __builtin_readcyclecounter(); rt0 = __builtin_amdgcn_s_memrealtime(); }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, half = lane >> 5;
    const int n0 = bx * BN, i0 = by * BM;
    const int mbeg = bz * a.m_per_split;
    const int mend = (mbeg + a.m_per_split < a.Mred) ? mbeg + a.m_per_split : a.Mred;
    const int Cs = a.g.Cs, taps_x = a.g.taps_x, Ws = a.g.Ws, step = a.g.step;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(a.dy, a.dy_bytes);
    const __amdgpu_buffer_rsrc_t rsB = U8 ? make_rsrc(reinterpret_cast<const float*>(a.g.src8), a.g.src_bytes)
                                          : make_rsrc(a.g.src + a.g.origin, a.g.src_bytes);

    // ---- per-thread constants: dy fragment offsets, gather column
    unsigned voffA[RA];
#pragma unroll
    for (int p = 0; p < RA; ++p) {
        int idx = tid + p * 256;
        int kl = idx / MC4, c4 = idx - kl * MC4;
        bool in_tile = NA4 % 256 == 0 || idx < NA4;
        if constexpr (SP) {                         // passes 2q, 2q + 1: rows 2 kl2, 2 kl2 + 1 of task tid + 256 q
            idx = tid + (p >> 1) * 256;
            const int kl2 = idx / MC4;
            c4 = idx - kl2 * MC4;
            kl = 2 * kl2 + (p & 1);
            in_tile = NPA % 256 == 0 || idx < NPA;
        }
        const int ko = i0 + c4 * 4;
        voffA[p] = (in_tile && ko < a.K_out) ? (unsigned)(kl * a.K_out + ko) << 2 : OOB;
    }
    // the reduction row (within a k-tile) that dy pass p of this thread covers
    auto a_row_of = [&](int p) { return SP ? 2 * ((tid + (p >> 1) * 256) / MC4) + (p & 1) : (tid + p * 256) / MC4; };
    const int b_c4 = tid % NC4, b_k0 = tid / NC4;
    const int r = n0 + b_c4 * CW;
    const int tap = r / Cs, ch = r - tap * Cs;
    const int cty = tap / taps_x, ctx = tap - cty * taps_x;
    unsigned cdelta = r < a.N ? (unsigned)(step * (cty * Ws + ctx) * Cs + ch - a.g.dmin) << 2 : OOB;
    if constexpr (U8) {
        const int khw = a.g.kh8 * a.g.kw8;
        const int pl = r / khw, rem = r - pl * khw;
        const int fy = rem / a.g.kw8, fx = rem - fy * a.g.kw8;
        cdelta = r < a.N ? (unsigned)(pl * a.g.plane + fy * Ws + fx) : OOB;
    }

    // ---- row producer state: thread t owns row t of every 256-row group
    int pm = mbeg + tid, pb, poy, pox;
    {
        const int t = pm / a.g.out_w;
        pox = pm - t * a.g.out_w;
        pb = t / a.g.out_h;
        poy = t - pb * a.g.out_h;
    }
    auto produce_rows = [&](int slot) {
        const int ry = poy * a.g.mul + a.g.add_y, rx = pox * a.g.mul + a.g.add_x;
        unsigned off = OOB, im = ~0u;
        if (U8 && pm < mend) {
            const int row = a.g.idx ? a.g.idx[pb] : pb;
            off = (unsigned)(row * a.g.img_bytes + ry * Ws + rx);
            im = 0;
        } else if (pm < mend) {
            off = (unsigned)(((pb * a.g.Hs + ry) * Ws + rx) * Cs - a.g.rmin) << 2;
            im = 0;
            if (HAS_PAD) im = tap_mask(ry, rx, a.g.Hs, Ws, a.g.taps_y, taps_x, step);
        }
        s_row[slot][tid] = make_uint2(off, im);
        // advance this thread's row by 256 (host-provided decomposition 256 = qb*out_h*out_w + qw*out_w + rw)
        pm += WG_ROWS;
        pb += a.adv_b; poy += a.adv_y; pox += a.adv_x;
        if (pox >= a.g.out_w) { pox -= a.g.out_w; ++poy; }
        if (poy >= a.g.out_h) { poy -= a.g.out_h; ++pb; }
    };

    // The workgroups of the first column tile also sum their dy rows per channel: the bias gradient's
    // partials ride along (4 RA vector adds per k-tile in 1 / (N / BN) of the workgroups).
    const bool do_bias = a.bias_part != nullptr && bx == 0;             // uniform
    float4 bsum[RA];
#pragma unroll
    for (int p = 0; p < RA; ++p) bsum[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 va[RA], vb[RB];
    unsigned vb8[RB];
    u32x2 vb8w[RB];
    auto issue_loads = [&](int tile) {              // tile index within the split
        const unsigned soffA = (unsigned)((mbeg + tile * BK) * a.K_out) << 2;
        const int grp = tile / TILES_PER_GROUP, tin = tile - grp * TILES_PER_GROUP;
        const uint2* rows = &s_row[grp & 1][tin * BK + (SP ? 2 * b_k0 : b_k0)];
        // (the buffer range check does not see soffset: U8's ragged last tile switches its missing dy rows off here)
        const int rows_left = mend - (mbeg + tile * BK);
#pragma unroll
        for (int p = 0; p < RA; ++p) {
            const bool row_ok = !U8 || rows_left >= BK || a_row_of(p) < rows_left;
            va[p] = buf_ld4s(rsA, row_ok ? voffA[p] : OOB, soffA);
        }
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            // split: passes 2q, 2q + 1 gather rows 2 (b_k0 + q KROWS), + 1
            const uint2 e = rows[SP ? (p >> 1) * 2 * KROWS + (p & 1) : p * KROWS];
            const unsigned off = e.x + cdelta;      // either term may be the OOB marker (sum stays >= OOB, < 2^32)
            if constexpr (W8) vb8w[p] = buf_ld2s(rsB, off, 0);
            else if constexpr (U8) vb8[p] = buf_ld1s(rsB, off, 0);
            else vb[p] = buf_ld4s(rsB, HAS_PAD ? mask_off(e.y, tap, off) : off, 0);
        }
    };
    auto pack4 = [&](const float4& v0, const float4& v1, char* d, int plane_bytes) {     // rows k, k + 1 -> three planes
        uint4 h, m, l;
        split_pair(v0.x, v1.x, h.x, m.x, l.x);
        split_pair(v0.y, v1.y, h.y, m.y, l.y);
        split_pair(v0.z, v1.z, h.z, m.z, l.z);
        split_pair(v0.w, v1.w, h.w, m.w, l.w);
        *reinterpret_cast<uint4*>(d) = h;
        *reinterpret_cast<uint4*>(d + plane_bytes) = m;
        *reinterpret_cast<uint4*>(d + 2 * plane_bytes) = l;
    };
    auto store_tiles = [&](int buf, bool fresh) {   // fresh: va holds a tile not stored before
        if constexpr (SP) {
            char* dS = sS + buf * STAGE;
#pragma unroll
            for (int q = 0; q < RA / 2; ++q) {
                const int t = tid + q * 256;
                if (NPA % 256 != 0 && t >= NPA) continue;
                pack4(va[2 * q], va[2 * q + 1], dS + t * 16, SPA);             // [kl2][c4 * 4] dwords, ld = BM
                if (do_bias && fresh) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int p = 2 * q + e;
                        bsum[p].x += va[p].x; bsum[p].y += va[p].y; bsum[p].z += va[p].z; bsum[p].w += va[p].w;
                    }
                }
            }
            char* dB = dS + 3 * SPA;
#pragma unroll
            for (int q = 0; q < RB / 2; ++q) {
                char* d = dB + ((b_k0 + q * KROWS) * BN + b_c4 * CW) * 4;
                if constexpr (W8) {                 // eight columns of the pair of rows: two 16-byte stores
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const float4 f0 = bytes_to_f4(vb8w[2 * q][hh]), f1 = bytes_to_f4(vb8w[2 * q + 1][hh]);
                        *reinterpret_cast<uint4*>(d + 16 * hh) = make_uint4(hi_pair(f0.x, f1.x), hi_pair(f0.y, f1.y),
                                                                            hi_pair(f0.z, f1.z), hi_pair(f0.w, f1.w));
                    }
                } else if constexpr (U8) {          // 0 .. 255 is exact in bf16: one plane
                    const float4 f0 = bytes_to_f4(vb8[2 * q]), f1 = bytes_to_f4(vb8[2 * q + 1]);
                    *reinterpret_cast<uint4*>(d) = make_uint4(hi_pair(f0.x, f1.x), hi_pair(f0.y, f1.y), hi_pair(f0.z, f1.z),
                                                              hi_pair(f0.w, f1.w));
                } else {
                    pack4(vb[2 * q], vb[2 * q + 1], d, SPB);
                }
            }
            return;
        }
        float* dA = sA + buf * A_SZ;
        float* dB = sB + buf * B_SZ;
#pragma unroll
        for (int p = 0; p < RA; ++p) {
            const int idx = tid + p * 256;
            if (NA4 % 256 != 0 && idx >= NA4) continue;
            *reinterpret_cast<float4*>(dA + idx * 4) = va[p];
            if (do_bias && fresh) { bsum[p].x += va[p].x; bsum[p].y += va[p].y; bsum[p].z += va[p].z; bsum[p].w += va[p].w; }
        }
#pragma unroll
        for (int p = 0; p < RB; ++p)
            *reinterpret_cast<float4*>(dB + (b_k0 + p * KROWS) * BN + b_c4 * 4) =
                U8 ? bytes_to_f4(vb8[p]) : vb[p];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    constexpr int G16 = TN * 2;                     // M16: 16-column groups per wave
    const int l15 = lane & 15, quad = lane >> 4;
    f32x4 acc16[G16];
#pragma unroll
    for (int gq = 0; gq < G16; ++gq)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc16[gq][v] = 0.f;

    // Row groups: group g (tiles g*T .. g*T+T-1) lives in slot g & 1.  Groups 0 and 1 are produced
    // up front; group g+2 is produced in the first iteration of group g+1's ... see loop.
    const int nk = U8 ? (mend - mbeg + BK - 1) / BK : (mend - mbeg) / BK;
    produce_rows(0);
    produce_rows(1);
    __syncthreads();
    issue_loads(0);
    store_tiles(nk & 1, true);                      // first tile's buffer chosen so that the loop ends on buffer 1
    __syncthreads();
    if (a.trace) tr1 = __builtin_readcyclecounter();
    // one k-tile with a compile-time buffer index (as in igemm_body: no vector address math between MFMAs)
    // Split products (round 5): the next tile's split (88 vector instructions + 6 LDS stores per thread) is dealt out
    // BETWEEN this tile's MFMAs -- it runs on the vector unit while the matrix pipe works, where the fp32-MFMA route
    // (which shares the vector unit's issue) wants it fenced behind them.  With the fence hipcc emitted, per wave and
    // k-tile: 12 reads, 9 MFMAs, 12 reads, 9 MFMAs, THEN the whole split, THEN the barrier.  Needs the tile loop without
    // a branch between the MFMAs and the stores: tiles 0 .. nk - 2 (LAST = false) always stage their successor.
    constexpr bool WIL = SP && ARL_WGRAD_INTERLEAVE && TM * TN == 1;   // (the 128 x 128 pair kernels run out of registers)
    auto k_tile = [&](auto buf_c, int kt, auto last_c) __attribute__((always_inline)) {
        constexpr int buf = decltype(buf_c)::value;
        constexpr bool LAST = decltype(last_c)::value;
        if (WIL ? !LAST : kt + 1 < nk) issue_loads(kt + 1);
        // tile kt+1 was the last reader of group (kt+1)/T when it is that group's last tile; the
        // slot is rewritten (group + 2) one iteration later, after this iteration's barrier.
        if (kt % TILES_PER_GROUP == 0 && kt >= TILES_PER_GROUP) produce_rows(((kt / TILES_PER_GROUP) + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (WIL && !LAST && !U8) {
            // Hand-made interleave (hipcc sinks the split behind the last MFMA under every scheduling hint tried:
            // profiles/r05/ring_conv_evidence.md): the MFMAs as volatile asm in the production order, and after MFMA n
            // the next few SLICES of the split of tile kt + 1 -- a task = one pack4 = 4 pairs x 4 stages of 3 / 3 / 3 / 2
            // vector instructions, then its three 16-byte LDS stores -- each slice pinned by an operand-tied empty asm.
            const unsigned* cA = reinterpret_cast<const unsigned*>(sS + buf * STAGE) + (half * 4) * BM + wm * TM * 32 + l31;
            const unsigned* cB = reinterpret_cast<const unsigned*>(sS + buf * STAGE + 3 * SPA) + (half * 4) * BN + wn * TN * 32 + l31;
            constexpr int NTA = RA / 2, NT = RA / 2 + RB / 2;       // split tasks of this thread: dy pairs, gathered pairs
            constexpr int ITEMS = NT * 19;                          // 16 slices + 3 stores per task
            constexpr int NMW = SP ? TM * TN * SPLIT * (BK / 16) : 1;   // (SP is true here; the other value keeps the
            constexpr int N0 = NMW >= 12 ? NMW / 6 : 0;                 //  discarded instantiations well-formed)
            constexpr int PER = (ITEMS + (NMW - N0) - 1) / (NMW - N0);  // the first MFMAs run while the tile's loads land
            float t0[NT][4], t1[NT][4], r0[NT][4], r1[NT][4];
            unsigned hh[NT][4], mm[NT][4], ll[NT][4];
            char* const dS = sS + (buf ^ 1) * STAGE;
            auto item = [&](const int it_) __attribute__((always_inline)) {     // (a constant after unrolling)
                const int T = it_ / 19, w = it_ - T * 19;
                const bool isA = T < NTA;
                const int q = isA ? T : T - NTA;
                const float4 v0 = isA ? va[2 * q] : vb[2 * q], v1 = isA ? va[2 * q + 1] : vb[2 * q + 1];
                if (w < 16) {
                    const int pr = w >> 2, stg = w & 3;
                    const float x0 = pr == 0 ? v0.x : pr == 1 ? v0.y : pr == 2 ? v0.z : v0.w;
                    const float x1 = pr == 0 ? v1.x : pr == 1 ? v1.y : pr == 2 ? v1.z : v1.w;
                    if (stg == 0) {
                        t0[T][pr] = __uint_as_float(__float_as_uint(x0) & HI16);
                        t1[T][pr] = __uint_as_float(__float_as_uint(x1) & HI16);
                        hh[T][pr] = hi_pair(x0, x1);
                        asm volatile("" : "+v"(t0[T][pr]), "+v"(t1[T][pr]), "+v"(hh[T][pr]));
                    } else if (stg == 1) {
                        r0[T][pr] = x0 - t0[T][pr];
                        r1[T][pr] = x1 - t1[T][pr];
                        t0[T][pr] = __uint_as_float(__float_as_uint(r0[T][pr]) & HI16);
                        asm volatile("" : "+v"(r0[T][pr]), "+v"(r1[T][pr]), "+v"(t0[T][pr]));
                    } else if (stg == 2) {
                        t1[T][pr] = __uint_as_float(__float_as_uint(r1[T][pr]) & HI16);
                        mm[T][pr] = hi_pair(r0[T][pr], r1[T][pr]);
                        r0[T][pr] = r0[T][pr] - t0[T][pr];
                        asm volatile("" : "+v"(t1[T][pr]), "+v"(mm[T][pr]), "+v"(r0[T][pr]));
                    } else {
                        r1[T][pr] = r1[T][pr] - t1[T][pr];
                        ll[T][pr] = hi_pair(r0[T][pr], r1[T][pr]);
                        asm volatile("" : "+v"(r1[T][pr]), "+v"(ll[T][pr]));
                    }
                } else {
                    const int pl = w - 16;
                    char* d = isA ? dS + (tid + q * 256) * 16 + pl * SPA
                                  : dS + 3 * SPA + ((b_k0 + q * KROWS) * BN + b_c4 * CW) * 4 + pl * SPB;
                    const bool live = !isA || NPA % 256 == 0 || tid + q * 256 < NPA;
                    const uint4 v = pl == 0 ? make_uint4(hh[T][0], hh[T][1], hh[T][2], hh[T][3])
                                  : pl == 1 ? make_uint4(mm[T][0], mm[T][1], mm[T][2], mm[T][3])
                                            : make_uint4(ll[T][0], ll[T][1], ll[T][2], ll[T][3]);
                    if (live) *reinterpret_cast<uint4*>(d) = v;
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            constexpr int NPR = SPLIT == 6 ? 6 : 9;                 // products, smallest first (split_products' order)
            constexpr int PA_[9] = {2, 1, 2, 0, 1, 2, 0, 1, 0}, PB_[9] = {2, 2, 1, 2, 1, 0, 1, 0, 0};
            constexpr int P0 = 9 - NPR;                             // (six products: the three smallest are dropped)
            static_assert(!SP || PER * (NMW - N0) >= ITEMS, "every slice has a slot");
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                u32x4 fa[TM][3], fb[TN][3];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const unsigned* q = cA + pl * (SPA / 4) + ks * 8 * BM + i * 32;
                        fa[i][pl] = u32x4{q[0], q[BM], q[2 * BM], q[3 * BM]};
                    }
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const unsigned* q = cB + pl * (SPB / 4) + ks * 8 * BN + j * 32;
                        fb[j][pl] = u32x4{q[0], q[BN], q[2 * BN], q[3 * BN]};
                    }
#pragma unroll
                for (int pi = 0; pi < NPR; ++pi)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i][j])
                                         : "v"(fa[i][PA_[P0 + pi]]), "v"(fb[j][PB_[P0 + pi]]));
                            const int n = ((ks * NPR + pi) * TM + i) * TN + j;       // this MFMA's number in the tile
#pragma unroll
                            for (int e = 0; e < PER; ++e) {
                                const int idx = (n - N0) * PER + e;
                                if (n >= N0 && idx < ITEMS) item(idx);
                            }
                        }
            }
            if (do_bias) {
#pragma unroll
                for (int p = 0; p < RA; ++p) { bsum[p].x += va[p].x; bsum[p].y += va[p].y; bsum[p].z += va[p].z; bsum[p].w += va[p].w; }
            }
            asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");       // the last asm MFMA's result before anything else reads it
            __syncthreads();
            return;
        }
        if constexpr (SP) {
            // fragment = column l31 (of its 32-wide tile), k octet 2 ks + half = pair rows 8 ks + 4 half .. + 3
            const unsigned* cA = reinterpret_cast<const unsigned*>(sS + buf * STAGE) + (half * 4) * BM + wm * TM * 32 + l31;
            const unsigned* cB = reinterpret_cast<const unsigned*>(sS + buf * STAGE + 3 * SPA) + (half * 4) * BN + wn * TN * 32 + l31;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                u32x4 fa[TM][3], fb[TN][3];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const unsigned* q = cA + pl * (SPA / 4) + ks * 8 * BM + i * 32;
                        fa[i][pl] = u32x4{q[0], q[BM], q[2 * BM], q[3 * BM]};
                    }
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int pl = 0; pl < PB; ++pl) {
                        const unsigned* q = cB + pl * (SPB / 4) + ks * 8 * BN + j * 32;
                        fb[j][pl] = u32x4{q[0], q[BN], q[2 * BN], q[3 * BN]};
                    }
                split_products<SPLIT, 3, PB, false, TM, TN>(fa, fb, acc);
            }
        } else if constexpr (M16) {
            const float* cA = sA + buf * A_SZ + (quad * 4) * BM + l15;
            const float* cB = sB + buf * B_SZ + (quad * 4) * BN + wn * TN * 32 + l15;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                float fa[4], fb[G16][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    fa[q] = cA[(ks * 16 + q) * BM];
#pragma unroll
                    for (int gq = 0; gq < G16; ++gq) fb[gq][q] = cB[(ks * 16 + q) * BN + gq * 16];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int gq = 0; gq < G16; ++gq)
                        acc16[gq] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[q], fb[gq][q], acc16[gq], 0, 0, 0);
            }
        } else {
        const float* cA = sA + buf * A_SZ + (half * 4) * BM + wm * TM * 32 + l31;
        const float* cB = sB + buf * B_SZ + (half * 4) * BN + wn * TN * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < BK / 8; ++ks) {
            float fa[TM][4], fb[TN][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i][q] = cA[(ks * 8 + q) * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j][q] = cB[(ks * 8 + q) * BN + j * 32];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][q], fb[j][q], acc[i][j], 0, 0, 0);
        }
        }
        if constexpr (WIL) {
            if constexpr (!LAST) {
                store_tiles(buf ^ 1, true);
                constexpr int NMW = TM * TN * (PB == 1 ? 3 : SPLIT) * (BK / 16);        // MFMAs per k-tile and wave
                constexpr int NVW = (RA / 2) * 44 + (RB / 2) * (U8 ? 12 : 44);           // the split's vector instructions
                constexpr int NWW = (RA / 2) * 3 + (RB / 2) * PB;                        // its LDS stores
                constexpr int VPMW = (NVW + NMW - 1) / NMW < 6 ? (NVW + NMW - 1) / NMW : 6;
                constexpr int WEVW = NMW / NWW > 0 ? NMW / NWW : 1;
                __builtin_amdgcn_sched_group_barrier(0x100, (BK / 16) * 4 * (TM * 3 + TN * PB), 0);   // every fragment read first
#pragma unroll
                for (int m = 0; m < NMW; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, VPMW, 0);
                    if (m % WEVW == WEVW - 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
                __syncthreads();
            }
        } else {
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < nk) {
                store_tiles(buf ^ 1, true);
                __syncthreads();
            }
        }
    };
    using WC0 = std::integral_constant<int, 0>;
    using WC1 = std::integral_constant<int, 1>;
    if constexpr (WIL) {        // tile j sits in buffer (j + nk) & 1: the last tile in buffer 1
        int kt = 0;
        if (!(nk & 1)) { k_tile(WC0{}, 0, std::false_type{}); kt = 1; }
        for (; kt + 1 < nk; kt += 2) {
            k_tile(WC1{}, kt, std::false_type{});
            k_tile(WC0{}, kt + 1, std::false_type{});
        }
        if (nk > 0) k_tile(WC1{}, nk - 1, std::true_type{});
    } else {   // an odd tile count peels its FIRST tile (from buffer 1); the rest is whole (buffer 0, buffer 1) pairs
        int kt = 0;
        if (nk & 1) { k_tile(WC1{}, 0, std::false_type{}); kt = 1; }
        for (; kt < nk; kt += 2) {
            k_tile(WC0{}, kt, std::false_type{});
            k_tile(WC1{}, kt + 1, std::false_type{});
        }
    }
    if (a.trace) tr2 = __builtin_readcyclecounter();
    if (do_bias) __syncthreads();                   // every wave is done with the tile buffers
    if (do_bias) {                                  // [BK][MC4] float4 in the (now idle) A buffer, summed in row order
        float4* red = reinterpret_cast<float4*>(sA);
#pragma unroll
        for (int p = 0; p < RA; ++p) {
            if constexpr (SP) {
                const int t = tid + (p >> 1) * 256;
                if (NPA % 256 == 0 || t < NPA) red[a_row_of(p) * MC4 + t % MC4] = bsum[p];
                continue;
            }
            const int idx = tid + p * 256;
            if (NA4 % 256 == 0 || idx < NA4) red[idx] = bsum[p];
        }
        __syncthreads();
        if (tid < MC4) {
            float4 t = red[tid];
            for (int kl = 1; kl < BK; ++kl) {
                const float4 v = red[kl * MC4 + tid];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            const int ko = i0 + tid * 4;
            if (ko < a.K_out) *reinterpret_cast<float4*>(a.bias_part + (int64_t)bz * a.K_out + ko) = t;
        }
    }

    float* out = a.part + (int64_t)bz * a.K_out * a.N;
    if constexpr (M16) {                            // D[row = 4 quad + v][col = l15] per 16-column group
#pragma unroll
        for (int gq = 0; gq < G16; ++gq) {
            const int col = n0 + wn * TN * 32 + gq * 16 + l15;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = i0 + 4 * quad + v;
                if (row < a.K_out && col < a.N) out[(int64_t)row * a.N + col] = U8 ? acc16[gq][v] * a.g.scale : acc16[gq][v];
            }
        }
    } else {
        if constexpr (U8) {                         // the pixel scale on the finished sums (see bytes_to_f4)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[i][j][v] *= a.g.scale;
        }
        store_tiles_rowmajor<TM, TN>(acc, out, a.K_out, a.N, i0 + wm * TM * 32, n0 + wn * TN * 32, lane, nullptr, 0);
    }
    if (a.trace && tid == 0) {                      // (plain launches only: the slot is the workgroup's grid index)
        unsigned long long* t = a.trace + ((size_t)(bz * gridDim.y + by) * gridDim.x + bx) * 8;
        t[0] = tr0; t[1] = tr1; t[2] = tr2; t[3] = __builtin_readcyclecounter();
        t[4] = rt0; t[5] = __builtin_amdgcn_s_memrealtime();
        t[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID
        t[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // XCC_ID
    }
}

template <int WGM, int WGN, int TM, int TN, int BK, bool HAS_PAD, bool M16 = false>
__global__ __launch_bounds__(256, (TM * TN > 1 ? 2 : 4)) void wgrad_fast_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    wgrad_fast_body<WGM, WGN, TM, TN, BK, HAS_PAD, M16>(a, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// weight gradient of a convolution whose input is the planar u8 observations (see wgrad_fast_body, U8)
template <int WGM, int WGN, int TM, int TN, int BK, bool M16>
__global__ __launch_bounds__(256, 4) void wgrad_u8_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    wgrad_fast_body<WGM, WGN, TM, TN, BK, false, M16, true>(a, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// bf16-split products (wgrad_fast_body, SPLIT), from f32 activations or (U8) the planar u8 observations
template <int WGM, int WGN, int TM, int TN, int BK, bool HAS_PAD, bool U8, int SPLIT, int MINW>
__global__ __launch_bounds__(256, MINW) void wgrad_split_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (a.xcd) {                                    // uniform
        const int gx = gridDim.x, gy = gridDim.y;
        const int t = xcd_chunk((bz * gy + by) * gx + bx, gx * gy * (int)gridDim.z);
        bx = t % gx;
        const int u = t / gx;
        by = u % gy; bz = u / gy;
    }
    wgrad_fast_body<WGM, WGN, TM, TN, BK, HAS_PAD, false, U8, SPLIT>(a, bx, by, bz, smem);
}

// One launch for a layer's data gradient AND weight gradient (independent of each other, both read dy):
// workgroups [0, n_ig) run data-gradient tiles, the rest weight-gradient tiles.  One ramp-up and one
// tail instead of two, and the dispatcher fills the CUs the first problem's last wave leaves idle.
template <int DWGM, int DWGN, int DTM, int DTN, int WWGM, int WWGN, int WTM, int WTN, int BK, bool HAS_PAD, int SPLIT = 0>
__global__ __launch_bounds__(256) void bwd_pair_kernel(const GemmArgs a, const WgradArgs w, const int dgx, const int dgy,
                                                       const int n_ig, const int wgx, const int wgy) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int id = blockIdx.x;
    if (a.xcd) id = xcd_chunk(id, (int)gridDim.x);     // uniform
    if (id < n_ig) {
        const int bx = id % dgx, t = id / dgx;          // the row tiles over one weight panel are neighbours
        igemm_body<DWGM, DWGN, DTM, DTN, BK, false, false, HAS_PAD, false, false, SPLIT>(a, bx, t % dgy, t / dgy, smem);
    } else {
        id -= n_ig;
        if (a.xcd) {                                    // ... and so are the row tiles over one panel of the layer's input
            const int by = id % wgy, t = id / wgy;
            wgrad_fast_body<WWGM, WWGN, WTM, WTN, BK, HAS_PAD, false, false, SPLIT>(w, t % wgx, by, t / wgx, smem);
        } else {
            const int bx = id % wgx, t = id / wgx;
            wgrad_fast_body<WWGM, WWGN, WTM, WTN, BK, HAS_PAD, false, false, SPLIT>(w, bx, t % wgy, t / wgy, smem);
        }
    }
}

template <int WGM, int WGN, int TM, int TN, int BK, bool B_KC, bool TAP_UNIFORM = false>
int launch_rowgather(const GemmArgs& a, int splits, hipStream_t s) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int A_SZ = BM * (BK + 4), B_SZ = B_KC ? BN * (BK + 4) : BK * BN;
    const size_t lds = (size_t)2 * (A_SZ + B_SZ) * sizeof(float);
    dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN, splits);
    hipLaunchKernelGGL((rowgather_gemm_kernel<WGM, WGN, TM, TN, BK, B_KC, TAP_UNIFORM>), grid, dim3(256), lds, s, a);
    return arl::check_launch("rowgather_gemm_kernel");
}

template <int WGM, int WGN, int TM, int TN, int BK>
int launch_wgrad(const WgradArgs& a, int splits, hipStream_t s) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    const size_t lds = (size_t)2 * BK * (BM + BN) * sizeof(float);
    dim3 grid((a.N + BN - 1) / BN, (a.K_out + BM - 1) / BM, splits);
    hipLaunchKernelGGL((wgrad_kernel<WGM, WGN, TM, TN, BK>), grid, dim3(256), lds, s, a);
    return arl::check_launch("wgrad_kernel");
}

constexpr bool lds_fits(int floats) { return floats * 4 <= 65536; }

template <typename K>
int allow_big_lds(K kernel, size_t lds) {           // > 64 KiB of dynamic LDS needs an explicit opt-in
    if (lds <= 65536) return 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { arl::set_error("hipFuncSetAttribute(LDS %zu): %s", lds, hipGetErrorString(e)); return (int)e; }
    return 0;
}

template <int WGM, int WGN, int TM, int TN, int BK, bool B_KC, bool N16 = false>
int launch_igemm(const GemmArgs& a, int splits, bool multi_tap, bool has_pad, hipStream_t s) {
    constexpr int BM = WGM * TM * (N16 ? 16 : 32), BN = N16 ? WGN * 16 : WGN * TN * 32;
    constexpr int A_SZ = BM * (BK + 4), B_SZ = B_KC ? BN * (BK + 4) : BK * (N16 ? BN + 4 : BN);
    const size_t lds = (size_t)2 * (A_SZ + B_SZ) * sizeof(float);
    dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN, a.n_par ? a.n_par : splits);
    int rc = 0;
#define ARL_IGEMM(MT, HP)                                                                                  \
    do {                                                                                                   \
        auto k = igemm_kernel<WGM, WGN, TM, TN, BK, B_KC, MT, HP, N16>;                                    \
        rc = allow_big_lds(k, lds);                                                                        \
        if (!rc) hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);                                        \
    } while (0)
    if (multi_tap && has_pad) ARL_IGEMM(true, true);
    else if (multi_tap) ARL_IGEMM(true, false);
    else if (has_pad) ARL_IGEMM(false, true);
    else ARL_IGEMM(false, false);
#undef ARL_IGEMM
    return rc ? rc : arl::check_launch("igemm_kernel");
}

// What one entry-point call carries down to its launches (set by the extern "C" function from its own arguments, for
// the duration of that call, on the calling thread: no state survives a call, none is shared between threads).
//   split   route of the fp32 contractions (arl_conv_geom::route): 0 = fp32 MFMA chain, 6 / 9 = bf16-split products
//   corun   an optimiser job (arl_corun_job) that the call's data-gradient launch may host in extra workgroups
struct CorunJob { arl::OptSeg seg; int blocks, host_blocks; };
static_assert(sizeof(CorunJob) <= sizeof(arl_corun_job), "arl_corun_job too small");
struct CallCtx { int split; const CorunJob* corun; bool corun_taken; };
extern thread_local CallCtx t_ctx;                 // (mfma_conv.hip)
#define g_split (t_ctx.split)
struct CallScope {
    explicit CallScope(int split, const arl_corun_job* job = nullptr) {
        t_ctx.split = split; t_ctx.corun = reinterpret_cast<const CorunJob*>(job); t_ctx.corun_taken = false;
    }
    ~CallScope() { t_ctx.corun = nullptr; }
};
// arl_conv_geom::route -> split mode (-1: not a route)
inline int split_of(const arl_conv_geom* g) {
    if (!g) return 9;
    return g->route == ARL_CONV_ROUTE_SPLIT9 ? 9 : g->route == ARL_CONV_ROUTE_FP32 ? 0 : g->route == ARL_CONV_ROUTE_SPLIT6 ? 6 : -1;
}
#define ARL_ROUTE_SCOPE(geom, job)                                                                         \
    ARL_REQUIRE(split_of(geom) >= 0, ARL_E_ARG, "conv route: ARL_CONV_ROUTE_SPLIT9, _FP32 or _SPLIT6");    \
    const CallScope call_scope_(split_of(geom), job)
// a pending optimiser job for a data-gradient launch to host?  (taken at most once per call)
inline bool corun_take(arl::OptSeg* c, dim3* grid) {
    if (!t_ctx.corun || t_ctx.corun_taken) return false;
    *c = t_ctx.corun->seg;
    c->co_blocks = t_ctx.corun->blocks < t_ctx.corun->host_blocks ? t_ctx.corun->blocks : t_ctx.corun->host_blocks;
    grid->x += (unsigned)c->co_blocks;
    t_ctx.corun_taken = true;
    return true;
}

template <int WGM, int WGN, int TM, int TN, int BK, bool B_KC, bool N16, int MINW>
int launch_igemm_occ(const GemmArgs& a, bool multi_tap, bool has_pad, hipStream_t s, int splits = 1) {
    constexpr int BM = WGM * TM * (N16 ? 16 : 32), BN = N16 ? WGN * 16 : WGN * TN * 32;
    constexpr int A_SZ = BM * (BK + 4), B_SZ = B_KC ? BN * (BK + 4) : BK * (N16 ? BN + 4 : BN);
    const size_t lds = (size_t)2 * (A_SZ + B_SZ) * sizeof(float);
    static_assert(lds_fits(2 * (A_SZ + B_SZ)), "<= 64 KiB of LDS");
    dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN, a.n_par ? a.n_par : splits);
    arl::OptSeg c = {};
    if constexpr (!B_KC) {                          // a data gradient hosts the call's optimiser job, if any
        if (!multi_tap && corun_take(&c, &grid)) {
            if (has_pad) hipLaunchKernelGGL((igemm_occ_kernel<WGM, WGN, TM, TN, BK, B_KC, false, true, N16, MINW, true>), grid, dim3(256), lds, s, a, c);
            else hipLaunchKernelGGL((igemm_occ_kernel<WGM, WGN, TM, TN, BK, B_KC, false, false, N16, MINW, true>), grid, dim3(256), lds, s, a, c);
            return arl::check_launch("igemm_occ_kernel (co-run)");
        }
    }
#define ARL_IGEMM_OCC(MT, HP) \
    hipLaunchKernelGGL((igemm_occ_kernel<WGM, WGN, TM, TN, BK, B_KC, MT, HP, N16, MINW>), grid, dim3(256), lds, s, a, c)
    if (multi_tap && has_pad) ARL_IGEMM_OCC(true, true);
    else if (multi_tap) ARL_IGEMM_OCC(true, false);
    else if (has_pad) ARL_IGEMM_OCC(false, true);
    else ARL_IGEMM_OCC(false, false);
#undef ARL_IGEMM_OCC
    return arl::check_launch("igemm_occ_kernel");
}

#ifdef ARL_NO_SPLIT6        // development builds: half the split kernels (mode 6 then runs the nine-product kernels)
#define ARL_BY_MODE(X6, X9) do { X9; } while (0)
#else
#define ARL_BY_MODE(X6, X9) do { if (g_split == 6) { X6; } else { X9; } } while (0)
#endif

// the launch of igemm_split_kernel: two LDS stages of (1 or 3) + 3 bf16 planes; a data gradient hosts the pending
// optimiser job like launch_igemm_occ
template <int WGM, int WGN, int TM, int TN, int BK, bool B_KC, bool U8, int MINW>
int launch_igemm_split(const GemmArgs& a, bool multi_tap, bool has_pad, hipStream_t s, int splits = 1) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    const size_t lds = (size_t)2 * ((U8 ? 1 : 3) * BM + 3 * BN) * BK * 2;
    const size_t lds_dir = (size_t)2 * 3 * BN * BK * 2;         // direct gathered operand: only the weights live in LDS
    (void)lds_dir;
    dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN, a.n_par ? a.n_par : splits);
    arl::OptSeg c = {};
    int rc = 0;
#define ARL_SPLIT_K(MT, HP, SPL, CO, AD)                                                                   \
    do {                                                                                                   \
        auto k = igemm_split_kernel<WGM, WGN, TM, TN, BK, B_KC, MT, HP, U8, SPL, MINW, CO, AD>;            \
        const size_t lds_k = (AD) ? lds_dir : lds;                                                         \
        rc = allow_big_lds(k, lds_k + (CO ? 64 : 0));                                                      \
        if (!rc) hipLaunchKernelGGL(k, grid, dim3(256), lds_k, s, a, c);                                   \
    } while (0)
// (one wave per row tile and one tap per k-tile: the gathered operand goes straight into the fragment registers)
#define ARL_SPLIT_PIN(MT, HP, SPL, CO)                                                                     \
    do {                                                                                                   \
        constexpr bool AD = WGN == 1 && (U8 || !(MT));                                                     \
        ARL_SPLIT_K(MT, HP, SPL, CO, AD);                                                                  \
    } while (0)
#define ARL_SPLIT_MODE(MT, HP, CO)                                                                         \
    do {                                                                                                   \
        ARL_BY_MODE(ARL_SPLIT_PIN(MT, HP, 6, CO), ARL_SPLIT_PIN(MT, HP, 9, CO));                           \
    } while (0)
    if constexpr (U8) {
        ARL_SPLIT_MODE(false, false, false);
    } else {
        if constexpr (!B_KC) {
            if (!multi_tap && corun_take(&c, &grid)) {
                if (has_pad) ARL_SPLIT_MODE(false, true, true); else ARL_SPLIT_MODE(false, false, true);
                return rc ? rc : arl::check_launch("igemm_split_kernel (co-run)");
            }
        }
        if (multi_tap && has_pad) ARL_SPLIT_MODE(true, true, false);
        else if (multi_tap) ARL_SPLIT_MODE(true, false, false);
        else if (has_pad) ARL_SPLIT_MODE(false, true, false);
        else ARL_SPLIT_MODE(false, false, false);
    }
#undef ARL_SPLIT_PIN
#undef ARL_SPLIT_MODE
#undef ARL_SPLIT_K
    return rc ? rc : arl::check_launch("igemm_split_kernel");
}

template <int WGM, int WGN, int TM, int TN, int BK, bool M16 = false>
int launch_wgrad_fast(const WgradArgs& a, int splits, bool has_pad, hipStream_t s) {
    constexpr int BM = M16 ? 16 : WGM * TM * 32, BN = WGN * TN * 32;
    const size_t lds = (size_t)2 * BK * (BM + BN) * sizeof(float);
    dim3 grid((a.N + BN - 1) / BN, (a.K_out + BM - 1) / BM, splits);
    int rc;
    if (has_pad) {
        auto k = wgrad_fast_kernel<WGM, WGN, TM, TN, BK, true, M16>;
        rc = allow_big_lds(k, lds + 4096);
        if (!rc) hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
    } else {
        auto k = wgrad_fast_kernel<WGM, WGN, TM, TN, BK, false, M16>;
        rc = allow_big_lds(k, lds + 4096);
        if (!rc) hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
    }
    return rc ? rc : arl::check_launch("wgrad_fast_kernel");
}

template <int WGM, int WGN, int TM, int TN, int BK, bool U8, int MINW>
int launch_wgrad_split(const WgradArgs& a, int splits, bool has_pad, hipStream_t s) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    const size_t lds = (size_t)2 * (3 * BM + (U8 ? 1 : 3) * BN) * BK * 2;
    dim3 grid((a.N + BN - 1) / BN, (a.K_out + BM - 1) / BM, splits);
    int rc = 0;
#define ARL_WSPLIT(HP, SPL)                                                                                \
    do {                                                                                                   \
        auto k = wgrad_split_kernel<WGM, WGN, TM, TN, BK, HP, U8, SPL, MINW>;                              \
        rc = allow_big_lds(k, lds + 4096);                                                                 \
        if (!rc) hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);                                        \
    } while (0)
    if constexpr (U8) {
        ARL_BY_MODE(ARL_WSPLIT(false, 6), ARL_WSPLIT(false, 9));
    } else if (has_pad) {
        ARL_BY_MODE(ARL_WSPLIT(true, 6), ARL_WSPLIT(true, 9));
    } else {
        ARL_BY_MODE(ARL_WSPLIT(false, 6), ARL_WSPLIT(false, 9));
    }
#undef ARL_WSPLIT
    return rc ? rc : arl::check_launch("wgrad_split_kernel");
}

constexpr int TARGET_WGS = 256;     // one workgroup per CU is already MFMA-bound (fp32 MFMA: 1 wave / SIMD)
constexpr int BKT = 32;             // k-tile of the skinny configurations (host-side split granularity)

extern unsigned long long* g_trace;   // arl_dev_conv_trace_buffer
extern bool g_force_generic;          // arl_dev_conv_force_generic: route every call to the generic kernels (tests)
extern int g_fwd_tile;                // arl_dev_fwd_tile: tile shape of the 33 .. 64-column forward kernels (-1: by size)

struct Geom {
    int64_t batch;
    int H, W, C, K, kh, kw, stride, pad_h, pad_w, Ho, Wo;
};

inline int check_geom(const arl_conv_geom* g, Geom* o) {
    if (!g || g->batch <= 0 || g->in_h <= 0 || g->in_w <= 0 || g->in_c <= 0 || g->out_c <= 0 || g->kh <= 0 ||
        g->kw <= 0 || g->stride <= 0 || g->pad_h < 0 || g->pad_w < 0) {
        arl::set_error("conv: bad geometry");
        return ARL_E_ARG;
    }
    if ((g->in_c & 3) || (g->out_c & 3)) {
        arl::set_error("conv: channel counts must be multiples of 4 (in %d, out %d)", g->in_c, g->out_c);
        return ARL_E_RANGE;
    }
    o->batch = g->batch; o->H = g->in_h; o->W = g->in_w; o->C = g->in_c; o->K = g->out_c;
    o->kh = g->kh; o->kw = g->kw; o->stride = g->stride; o->pad_h = g->pad_h; o->pad_w = g->pad_w;
    o->Ho = (g->in_h + 2 * g->pad_h - g->kh) / g->stride + 1;
    o->Wo = (g->in_w + 2 * g->pad_w - g->kw) / g->stride + 1;
    const int64_t lim = (int64_t)OOB / 4;       // elements: every tensor must stay below the OOB byte offset
    if (o->Ho <= 0 || o->Wo <= 0 || g->batch * (int64_t)o->Ho * o->Wo * g->out_c >= lim ||
        g->batch * (int64_t)g->in_h * g->in_w * g->in_c >= lim ||
        (int64_t)g->out_c * g->kh * g->kw * g->in_c >= lim) {
        arl::set_error("conv: tensor larger than the 2 GiB the 32-bit buffer offsets address");
        return ARL_E_RANGE;
    }
    return 0;
}

inline int round_up(int x, int q) { return (x + q - 1) / q * q; }

// ceil(2^32 / d) if floor(n * that / 2^32) == n / d for every 0 <= n < rows (needs rows * d < 2^32), else 0
inline unsigned div_magic(int64_t rows, int d) {
    if (d <= 1 || rows * (int64_t)d >= ((int64_t)1 << 32)) return 0;
    return (unsigned)((((uint64_t)1 << 32) + (uint64_t)d - 1) / (uint64_t)d);
}

// 33 .. 64 output columns, the default: 32x64 tiles -- each wave two 16-row groups of one 16-column stripe
// (v_mfma_f32_16x16x4_f32), 32-deep k-tiles, two LDS stages (28 KB), compiled for five waves per SIMD.  Small tiles
// spread the rows evenly (1 728 tiles at the PPO minibatch: 7 on the busiest CU against 6.75 on average, where 864
// tiles of 64 rows leave it 4 against 3.375) and five or six resident workgroups per CU cover each other's barriers,
// prologues and epilogues.  Measured, 20 launches per hipGraph, conv 2 / conv 3 forward at 512 images: 36.6 / 40.1 us
// (64x64: 44.7 / 47.9, 112x64: 40.6 / 42.0); at 256: 22.4 / 24.1 (25.0 / 27.6, 25.4 / 26.7); at 128: 14.2 / 15.7
// (15.6 / 17.3, 22.9 / 25.2); 16-deep k-tiles at 6-8 waves per SIMD and 48-row tiles were slower everywhere
// (profiles/r02/tile_probe.txt).
template <bool B_KC>
int launch_n64(const GemmArgs& a, bool multi_tap, bool has_pad, hipStream_t s) {
    return launch_igemm_occ<1, 4, 2, 1, 32, B_KC, true, 5>(a, multi_tap, has_pad, s);
}

// split the reduction so that tiles * splits ~ TARGET_WGS, each split a multiple of BKT
inline void plan_split(int tiles, int red, int* splits, int* per, int want = TARGET_WGS) {
    int s = tiles >= want ? 1 : want / tiles;
    const int max_s = (red + 4 * BKT - 1) / (4 * BKT);          // at least 4 k-tiles per split
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    *per = round_up((red + s - 1) / s, BKT);
    *splits = (red + *per - 1) / *per;
}

// Fast-path launch descriptions, so that a layer's data and weight gradient can share one launch
// (arl_conv2d_bwd_pair).  cfg: data gradient 0 = <4,1,1,1>, 1 = <2,2,1,1>, 2 = <2,2,2,2>;
// weight gradient 0 = <1,4,1,1>, 1 = <2,2,1,1>, 2 = <2,2,2,2>.
struct DgradPlan { GemmArgs a; bool fast, has_pad; int cfg; };
struct WgradPlan { WgradArgs a; bool fast, has_pad; int cfg, splits; int64_t total; };

template <int DWGM, int DWGN, int DTM, int DTN, int WWGM, int WWGN, int WTM, int WTN, int BK = 32>
int launch_pair(const DgradPlan& d, const WgradPlan& w, bool has_pad, hipStream_t s) {
    constexpr int DBM = DWGM * DTM * 32, DBN = DWGN * DTN * 32, WBM = WWGM * WTM * 32, WBN = WWGN * WTN * 32;
    const size_t lds_d = (size_t)2 * (DBM * (BK + 4) + BK * DBN) * sizeof(float);
    const size_t lds_w = (size_t)2 * BK * (WBM + WBN) * sizeof(float);
    const size_t lds = lds_d > lds_w ? lds_d : lds_w;
    const int dgx = (d.a.M + DBM - 1) / DBM, dgy = (d.a.N + DBN - 1) / DBN, dgz = d.a.n_par ? d.a.n_par : 1;
    const int wgx = (w.a.N + WBN - 1) / WBN, wgy = (w.a.K_out + WBM - 1) / WBM, wgz = w.splits;
    const int n_ig = dgx * dgy * dgz, n_wg = wgx * wgy * wgz;
    int rc;
    if (g_split) {
        const size_t lds_s = (size_t)2 * 3 * ((DBM + DBN) > (WBM + WBN) ? (DBM + DBN) : (WBM + WBN)) * BK * 2;
#define ARL_PSPLIT(HP, SPL)                                                                                \
    do {                                                                                                   \
        auto k = bwd_pair_kernel<DWGM, DWGN, DTM, DTN, WWGM, WWGN, WTM, WTN, BK, HP, SPL>;                 \
        rc = allow_big_lds(k, lds_s + 4096);                                                               \
        if (!rc) hipLaunchKernelGGL(k, dim3(n_ig + n_wg), dim3(256), lds_s, s, d.a, w.a, dgx, dgy, n_ig, wgx, wgy); \
    } while (0)
        if (has_pad) ARL_BY_MODE(ARL_PSPLIT(true, 6), ARL_PSPLIT(true, 9));
        else ARL_BY_MODE(ARL_PSPLIT(false, 6), ARL_PSPLIT(false, 9));
#undef ARL_PSPLIT
    } else if (has_pad) {
        auto k = bwd_pair_kernel<DWGM, DWGN, DTM, DTN, WWGM, WWGN, WTM, WTN, BK, true>;
        rc = allow_big_lds(k, lds + 4096);
        if (!rc) hipLaunchKernelGGL(k, dim3(n_ig + n_wg), dim3(256), lds, s, d.a, w.a, dgx, dgy, n_ig, wgx, wgy);
    } else {
        auto k = bwd_pair_kernel<DWGM, DWGN, DTM, DTN, WWGM, WWGN, WTM, WTN, BK, false>;
        rc = allow_big_lds(k, lds + 4096);
        if (!rc) hipLaunchKernelGGL(k, dim3(n_ig + n_wg), dim3(256), lds, s, d.a, w.a, dgx, dgy, n_ig, wgx, wgy);
    }
    return rc ? rc : arl::check_launch("bwd_pair_kernel");
}


// img_conv.hip: the image-stationary kernels (>= 0: launched / error code; -1: not their geometry)
int launch_conv1_img(const unsigned char* obs, int64_t obs_rows, const int32_t* idx, float scale, const float* w, const float* bias, float* y,
                     int64_t batch, int C, int H, int W, int K, int kh, int kw, int stride, int Ho, int Wo, int relu,
                     hipStream_t s);

// ---- the launcher instantiations, dealt out to translation units (mfma_conv_p<k>.hip define ARL_CONV_PART = k and hold the
// definitions of part k; every other unit sees them as extern templates) so that hipcc builds them side by side
#define ARL_GA const GemmArgs&, bool, bool, hipStream_t, int
#define ARL_P1(T) \
    T int launch_igemm_split<4, 1, 2, 1, 32, true, true, 2>(ARL_GA); \
    T int launch_igemm_split<4, 1, 1, 2, 32, true, false, 2>(ARL_GA); \
    T int launch_igemm_split<4, 1, 1, 1, 32, true, false, 2>(ARL_GA);
#define ARL_P2(T) \
    T int launch_igemm_split<2, 2, 2, 2, 32, true, false, 1>(ARL_GA); \
    T int launch_igemm_split<2, 2, 1, 1, 32, true, false, 3>(ARL_GA); \
    T int launch_igemm_split<2, 2, 1, 1, 32, false, false, 3>(ARL_GA);
#define ARL_P3(T) \
    T int launch_igemm_split<4, 1, 1, 2, 32, false, false, 2>(ARL_GA); \
    T int launch_igemm_split<4, 1, 1, 1, 32, false, false, 2>(ARL_GA); \
    T int launch_igemm_split<2, 2, 2, 2, 32, false, false, 1>(ARL_GA);
#define ARL_P4(T) \
    T int launch_wgrad_split<2, 2, 2, 2, 32, false, 1>(const WgradArgs&, int, bool, hipStream_t); \
    T int launch_wgrad_split<2, 2, 1, 1, 32, false, 2>(const WgradArgs&, int, bool, hipStream_t); \
    T int launch_wgrad_split<1, 4, 1, 1, 32, false, 2>(const WgradArgs&, int, bool, hipStream_t); \
    T int launch_wgrad_split<1, 4, 1, 1, 32, true, 2>(const WgradArgs&, int, bool, hipStream_t); \
    T int launch_wgrad_split<1, 4, 1, 2, 32, true, 2>(const WgradArgs&, int, bool, hipStream_t);
#define ARL_P5(T) \
    T int launch_pair<2, 2, 2, 2, 2, 2, 2, 2, 32>(const DgradPlan&, const WgradPlan&, bool, hipStream_t); \
    T int launch_pair<2, 2, 2, 2, 2, 2, 2, 2, 16>(const DgradPlan&, const WgradPlan&, bool, hipStream_t);
#define ARL_P6(T) \
    T int launch_igemm<2, 2, 1, 1, 32, true, false>(const GemmArgs&, int, bool, bool, hipStream_t); \
    T int launch_igemm<2, 2, 1, 1, 32, false, false>(const GemmArgs&, int, bool, bool, hipStream_t); \
    T int launch_igemm<2, 2, 2, 2, 32, true, false>(const GemmArgs&, int, bool, bool, hipStream_t); \
    T int launch_igemm<2, 2, 2, 2, 32, false, false>(const GemmArgs&, int, bool, bool, hipStream_t); \
    T int launch_igemm<4, 1, 2, 1, 16, true, true>(const GemmArgs&, int, bool, bool, hipStream_t); \
    T int launch_igemm<4, 1, 2, 1, 16, false, true>(const GemmArgs&, int, bool, bool, hipStream_t); \
    T int launch_igemm<4, 1, 1, 1, 32, true, false>(const GemmArgs&, int, bool, bool, hipStream_t); \
    T int launch_igemm<4, 1, 1, 1, 16, true, false>(const GemmArgs&, int, bool, bool, hipStream_t); \
    T int launch_igemm<4, 1, 1, 1, 16, false, false>(const GemmArgs&, int, bool, bool, hipStream_t);
#define ARL_P7(T) \
    T int launch_igemm_occ<2, 2, 2, 1, 32, false, true, 5>(ARL_GA); \
    T int launch_igemm_occ<1, 4, 2, 1, 32, true, true, 5>(ARL_GA); \
    T int launch_igemm_occ<1, 4, 2, 1, 32, false, true, 5>(ARL_GA); \
    T int launch_wgrad_fast<2, 2, 2, 2, 32, false>(const WgradArgs&, int, bool, hipStream_t); \
    T int launch_wgrad_fast<2, 2, 1, 1, 32, false>(const WgradArgs&, int, bool, hipStream_t); \
    T int launch_wgrad_fast<1, 4, 1, 1, 32, false>(const WgradArgs&, int, bool, hipStream_t); \
    T int launch_wgrad_fast<1, 4, 1, 1, 32, true>(const WgradArgs&, int, bool, hipStream_t);
#ifndef ARL_CONV_PART
#define ARL_CONV_PART 0
#endif
#define ARL_T_DEF template
#define ARL_T_EXT extern template
#if ARL_CONV_PART == 1
ARL_P1(ARL_T_DEF)
#else
ARL_P1(ARL_T_EXT)
#endif
#if ARL_CONV_PART == 2
ARL_P2(ARL_T_DEF)
#else
ARL_P2(ARL_T_EXT)
#endif
#if ARL_CONV_PART == 3
ARL_P3(ARL_T_DEF)
#else
ARL_P3(ARL_T_EXT)
#endif
#if ARL_CONV_PART == 4
ARL_P4(ARL_T_DEF)
#else
ARL_P4(ARL_T_EXT)
#endif
#if ARL_CONV_PART == 5
ARL_P5(ARL_T_DEF)
#else
ARL_P5(ARL_T_EXT)
#endif
#if ARL_CONV_PART == 6
ARL_P6(ARL_T_DEF)
#else
ARL_P6(ARL_T_EXT)
#endif
#if ARL_CONV_PART == 7
ARL_P7(ARL_T_DEF)
#else
ARL_P7(ARL_T_EXT)
#endif

}  // namespace arlc
