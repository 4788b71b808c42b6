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

// The kernels below are written for ONE target: the hand-scheduled weight-gradient tile issues v_mfma_f32_32x32x16_bf16 as
// volatile inline assembly under its gfx950 mnemonic with the wait states that pipe needs (LLVM's hazard recogniser does
// not look inside), and every tile shape, LDS image and register budget is sized for CDNA4.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "accel_rl_amd's MFMA kernels are written for gfx950 (MI355X) only: build with --offload-arch=gfx950"
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
    int cls;        // k-contiguous weights of a data gradient (arl_conv2d_dgrad_weights): elements between the parity
                    // classes' matrices (class z's rows start at w + z * cls); 0 otherwise
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
                 "s"(a.b.si), "s"(a.b.kw), "s"(a.b.c), "s"(a.b.cls));
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
// SPLIT = 1 (the labelled bf16 route, arl_conv_geom.route = ARL_CONV_ROUTE_BF16): ONE plane per operand, rounded to
// nearest even (v_cvt_pk_bf16_f32) -- a truncated piece is only right as the first term of an exact sum
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned rne_pair(float x0, float x1) {
    const f32x2_t v = {x0, x1};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// planes of an fp32 operand under a split mode, and its pieces (NP == 1: h alone, rounded; m and l untouched)
constexpr int planes_of(int split) { return split == 1 ? 1 : 3; }
template <int NP> __device__ __forceinline__ void split_pair_n(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    if constexpr (NP == 1) h = rne_pair(x0, x1);
    else split_pair(x0, x1, h, m, l);
}

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

// arguments of the weight-gradient kernels (generic: mfma_generic.h; scalar-addressed / split: mfma_wgrad.h)
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
// SPLIT = 6: the terms below 2^-24 |x y| (m l, l m, l l) are dropped; SPLIT = 1: plain bf16 operands (rounded, one
// product: NOT an fp32 contraction -- a labelled option, never a default).  The split happens between the global load
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

}  // namespace arlc
