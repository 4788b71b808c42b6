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

#include "arl_common.h"

#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
};

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

// out[i] = act(sum_z part[z][i] + bias[i % n_bias]) with a fixed summation order; float4 lanes.
__global__ __launch_bounds__(256) void fold_splits_kernel(const float4* __restrict__ part, int splits,
                                                          int64_t total4, const float4* __restrict__ bias,
                                                          int bias4, int relu, float4* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
        float4 s = part[i];
        for (int z = 1; z < splits; ++z) {
            const float4 v = part[(int64_t)z * total4 + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (bias) {
            const float4 b = bias[i % bias4];
            s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
        }
        if (relu) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
        out[i] = s;
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

int launch_fold(const float* part, int splits, int64_t total, const float* bias, int n_bias, int relu,
                float* out, hipStream_t s) {
    const int64_t total4 = total >> 2;
    hipLaunchKernelGGL(fold_splits_kernel, dim3(arl::stream_grid(total4, 256)), dim3(256), 0, s,
                       (const float4*)part, splits, total4, (const float4*)bias, n_bias >> 2, relu, (float4*)out);
    return arl::check_launch("fold_splits_kernel");
}

constexpr int TARGET_WGS = 256;     // one workgroup per CU is already MFMA-bound (fp32 MFMA: 1 wave / SIMD)
constexpr int BKT = 32;             // k-tile of the skinny configurations (host-side split granularity)

int tuning_bk() {                   // ARL_CONV_BK=16|32 overrides the k-tile (tuning aid)
    static int bk = [] { const char* e = getenv("ARL_CONV_BK"); return e ? atoi(e) : 0; }();
    return bk;
}
int tuning_tile() {                 // ARL_CONV_TILE=1: halve the row tile of the skinny configurations (tuning aid)
    static int t = [] { const char* e = getenv("ARL_CONV_TILE"); return e ? atoi(e) : 0; }();
    return t;
}

struct Geom {
    int64_t batch;
    int H, W, C, K, kh, kw, stride, pad_h, pad_w, Ho, Wo;
};

int check_geom(const arl_conv_geom* g, Geom* o) {
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

int round_up(int x, int q) { return (x + q - 1) / q * q; }

// split the reduction so that tiles * splits ~ TARGET_WGS, each split a multiple of BKT
void plan_split(int tiles, int red, int* splits, int* per) {
    int s = tiles >= TARGET_WGS ? 1 : TARGET_WGS / tiles;
    const int max_s = (red + 4 * BKT - 1) / (4 * BKT);          // at least 4 k-tiles per split
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    *per = round_up((red + s - 1) / s, BKT);
    *splits = (red + *per - 1) / *per;
}

}  // namespace

extern "C" int64_t arl_conv_workspace_bytes(void) { return (int64_t)64 << 20; }

extern "C" int arl_conv2d_fwd(const float* x, const float* w, const float* bias_or_null, float* y,
                              const arl_conv_geom* geom, int32_t relu, void* workspace, void* stream) {
    ARL_REQUIRE(x && w && y && workspace, ARL_E_ARG, "null pointer");
    Geom g;
    int rc = check_geom(geom, &g);
    if (rc) return rc;
    ARL_REQUIRE(arl::aligned16(x) && arl::aligned16(w) && arl::aligned16(y) && arl::aligned16(workspace) &&
                    (!bias_or_null || arl::aligned16(bias_or_null)), ARL_E_ALIGN, "16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    GemmArgs a = {};
    a.g.src = x; a.g.Hs = g.H; a.g.Ws = g.W; a.g.Cs = g.C; a.g.out_h = g.Ho; a.g.out_w = g.Wo;
    a.g.mul = g.stride; a.g.add_y = -g.pad_h; a.g.add_x = -g.pad_w; a.g.taps_x = g.kw; a.g.step = 1;
    a.M = (int)(g.batch * g.Ho * g.Wo); a.N = g.K; a.K = g.kh * g.kw * g.C;
    a.g.src_bytes = (unsigned)(g.batch * g.H * g.W * g.C * 4);
    a.b.w = w; a.b.ld = a.K; a.b.w_bytes = (unsigned)((int64_t)a.N * a.K * 4);
    a.o.dense = 1;
    int splits = 1, per = round_up(a.K, BKT);
    const bool small = a.N >= 128 && (int64_t)a.M * a.N <= (int64_t)1 << 20;     // dense layers: split K
    if (small) plan_split(((a.M + 63) / 64) * ((a.N + 63) / 64), a.K, &splits, &per);
    a.k_per_split = per;
    if (splits > 1) {
        ARL_REQUIRE((int64_t)splits * a.M * a.N * 4 <= arl_conv_workspace_bytes(), ARL_E_RANGE, "workspace too small");
        a.o.out = (float*)workspace; a.split_stride = (int64_t)a.M * a.N;
    } else {
        a.o.out = y; a.o.bias = bias_or_null; a.o.relu = relu;
    }
    const bool bk16 = tuning_bk() == 16;
    const int tt = tuning_tile();
    if (a.N <= 32 && tt == 1) rc = bk16 ? launch_rowgather<4, 1, 1, 1, 16, true>(a, splits, s) : launch_rowgather<4, 1, 1, 1, 32, true>(a, splits, s);
    else if (a.N <= 32) rc = bk16 ? launch_rowgather<4, 1, 2, 1, 16, true>(a, splits, s) : launch_rowgather<4, 1, 2, 1, 32, true>(a, splits, s);
    else if (a.N <= 64 && tt == 1) rc = bk16 ? launch_rowgather<2, 2, 1, 1, 16, true>(a, splits, s) : launch_rowgather<2, 2, 1, 1, 32, true>(a, splits, s);
    else if (a.N <= 64 && tt == 2) rc = bk16 ? launch_rowgather<4, 1, 1, 2, 16, true>(a, splits, s) : launch_rowgather<4, 1, 1, 2, 32, true>(a, splits, s);
    else if (a.N <= 64) rc = bk16 ? launch_rowgather<2, 2, 2, 1, 16, true>(a, splits, s) : launch_rowgather<2, 2, 2, 1, 32, true>(a, splits, s);
    else if (small) rc = bk16 ? launch_rowgather<2, 2, 1, 1, 16, true>(a, splits, s) : launch_rowgather<2, 2, 1, 1, 32, true>(a, splits, s);
    else rc = launch_rowgather<2, 2, 2, 2, 16, true>(a, splits, s);
    if (rc || splits == 1) return rc;
    return launch_fold((const float*)workspace, splits, (int64_t)a.M * a.N, bias_or_null, a.N, relu, y, s);
}

extern "C" int arl_conv2d_bwd_data(const float* dy, const float* w, const float* mask_or_null, float* dx,
                                   const arl_conv_geom* geom, void* stream) {
    ARL_REQUIRE(dy && w && dx, ARL_E_ARG, "null pointer");
    Geom g;
    int rc = check_geom(geom, &g);
    if (rc) return rc;
    ARL_REQUIRE(g.kh % g.stride == 0 && g.kw % g.stride == 0, ARL_E_RANGE,
                "data gradient needs kernel size divisible by stride");
    ARL_REQUIRE(arl::aligned16(dy) && arl::aligned16(w) && arl::aligned16(dx) &&
                    (!mask_or_null || arl::aligned16(mask_or_null)), ARL_E_ALIGN, "16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    const int st = g.stride;
    for (int ph = 0; ph < st && ph < g.H; ++ph) {
        for (int pw = 0; pw < st && pw < g.W; ++pw) {
            // input pixels (h, w) = (st*oy + ph, st*ox + pw); taps i = i0 + st*ti reach
            // output row (h + pad - i) / st = oy + (ph + pad - i0)/st - ti
            const int i0 = (ph + g.pad_h) % st, j0 = (pw + g.pad_w) % st;
            GemmArgs a = {};
            a.g.src = dy; a.g.Hs = g.Ho; a.g.Ws = g.Wo; a.g.Cs = g.K;
            a.g.out_h = (g.H - ph + st - 1) / st; a.g.out_w = (g.W - pw + st - 1) / st;
            a.g.mul = 1; a.g.add_y = (ph + g.pad_h - i0) / st; a.g.add_x = (pw + g.pad_w - j0) / st;
            a.g.taps_x = g.kw / st; a.g.step = -1;
            a.M = (int)(g.batch * a.g.out_h * a.g.out_w); a.N = g.C; a.K = (g.kh / st) * (g.kw / st) * g.K;
            a.g.src_bytes = (unsigned)(g.batch * g.Ho * g.Wo * g.K * 4);
            a.b.w_bytes = (unsigned)((int64_t)g.K * g.kh * g.kw * g.C * 4);
            a.b.w = w; a.b.ld = g.kh * g.kw * g.C; a.b.kc = g.K; a.b.taps_x = g.kw / st;
            a.b.i0 = i0; a.b.j0 = j0; a.b.si = st; a.b.kw = g.kw; a.b.c = g.C;
            a.o.out = dx; a.o.mask = mask_or_null; a.o.dense = (st == 1);
            a.o.OH = g.H; a.o.OW = g.W; a.o.omul = st; a.o.oadd_y = ph; a.o.oadd_x = pw;
            a.k_per_split = round_up(a.K, BKT);
            const bool bk16 = tuning_bk() == 16;
            const bool uni = g.K % 32 == 0;                 // a k-tile never straddles two filter taps
            const int tt = tuning_tile();
            if (a.N <= 32 && uni && tt == 1) {
                rc = bk16 ? launch_rowgather<4, 1, 1, 1, 16, false, true>(a, 1, s) : launch_rowgather<4, 1, 1, 1, 32, false, true>(a, 1, s);
            } else if (a.N <= 64 && a.N > 32 && uni && tt == 1) {
                rc = bk16 ? launch_rowgather<2, 2, 1, 1, 16, false, true>(a, 1, s) : launch_rowgather<2, 2, 1, 1, 32, false, true>(a, 1, s);
            } else if (a.N <= 32) {
                if (!uni) rc = launch_rowgather<4, 1, 2, 1, 16, false, false>(a, 1, s);
                else rc = bk16 ? launch_rowgather<4, 1, 2, 1, 16, false, true>(a, 1, s) : launch_rowgather<4, 1, 2, 1, 32, false, true>(a, 1, s);
            } else if (a.N <= 64) {
                if (!uni) rc = launch_rowgather<2, 2, 2, 1, 16, false, false>(a, 1, s);
                else rc = bk16 ? launch_rowgather<2, 2, 2, 1, 16, false, true>(a, 1, s) : launch_rowgather<2, 2, 2, 1, 32, false, true>(a, 1, s);
            } else {
                rc = uni ? launch_rowgather<2, 2, 2, 2, 16, false, true>(a, 1, s) : launch_rowgather<2, 2, 2, 2, 16, false, false>(a, 1, s);
            }
            if (rc) return rc;
        }
    }
    return 0;
}

extern "C" int arl_conv2d_bwd_weight(const float* dy, const float* x, float* dw, const arl_conv_geom* geom,
                                     void* workspace, void* stream) {
    ARL_REQUIRE(dy && x && dw && workspace, ARL_E_ARG, "null pointer");
    Geom g;
    int rc = check_geom(geom, &g);
    if (rc) return rc;
    ARL_REQUIRE(arl::aligned16(dy) && arl::aligned16(x) && arl::aligned16(dw) && arl::aligned16(workspace),
                ARL_E_ALIGN, "16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    WgradArgs a = {};
    a.dy = dy;
    a.g.src = x; a.g.Hs = g.H; a.g.Ws = g.W; a.g.Cs = g.C; a.g.out_h = g.Ho; a.g.out_w = g.Wo;
    a.g.mul = g.stride; a.g.add_y = -g.pad_h; a.g.add_x = -g.pad_w; a.g.taps_x = g.kw; a.g.step = 1;
    a.K_out = g.K; a.N = g.kh * g.kw * g.C; a.Mred = (int)(g.batch * g.Ho * g.Wo);
    a.g.src_bytes = (unsigned)(g.batch * g.H * g.W * g.C * 4);
    a.dy_bytes = (unsigned)((int64_t)a.Mred * g.K * 4);
    int bm, bn;
    if (g.K <= 32) { bm = 32; bn = 128; }
    else if (g.K <= 64) { bm = 64; bn = 64; }
    else { bm = 128; bn = 128; }
    const int tiles = ((a.K_out + bm - 1) / bm) * ((a.N + bn - 1) / bn);
    int splits, per;
    plan_split(tiles, a.Mred, &splits, &per);
    a.m_per_split = per;
    const int64_t total = (int64_t)a.K_out * a.N;
    if (splits > 1) {
        ARL_REQUIRE((int64_t)splits * total * 4 <= arl_conv_workspace_bytes(), ARL_E_RANGE, "workspace too small");
        a.part = (float*)workspace;
    } else {
        a.part = dw;
    }
    const bool bk16 = tuning_bk() == 16;
    if (g.K <= 32) rc = bk16 ? launch_wgrad<1, 4, 1, 1, 16>(a, splits, s) : launch_wgrad<1, 4, 1, 1, 32>(a, splits, s);
    else if (g.K <= 64) rc = bk16 ? launch_wgrad<2, 2, 1, 1, 16>(a, splits, s) : launch_wgrad<2, 2, 1, 1, 32>(a, splits, s);
    else rc = launch_wgrad<2, 2, 2, 2, 16>(a, splits, s);        // 128x128 at BK=32 would exceed 64 KB of LDS
    if (rc || splits == 1) return rc;
    return launch_fold((const float*)workspace, splits, total, nullptr, 4, 0, dw, s);
}
