// Generic fallbacks of the MFMA contractions (any channel count multiple of 4, any K): rowgather_gemm_kernel (forward /
// data gradient) and wgrad_kernel -- used when a k-tile would straddle filter taps, and by the parity tests
// (arl_dev_conv_force_generic).  Part of mfma_conv_impl.h.
#pragma once
#include "mfma_common.h"

namespace arlc {

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

}  // namespace arlc
