// Scalar-addressed weight gradient (wgrad_fast_body and its kernels: fp32 MFMA chain, u8 input, bf16-split products with the
// hand-dealt split).  Part of mfma_conv_impl.h.
#pragma once
#include "mfma_common.h"

namespace arlc {

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
    constexpr int PA = planes_of(SPLIT), PB = U8 ? 1 : planes_of(SPLIT);
    constexpr int SPA = BK * BM * 2, SPB = BK * BN * 2, STAGE = PA * SPA + PB * SPB;  // bytes per plane / stage
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
    auto pack4 = [&](const float4& v0, const float4& v1, char* d, int plane_bytes) {     // rows k, k + 1 -> PA planes
        uint4 h, m, l;
        split_pair_n<PA>(v0.x, v1.x, h.x, m.x, l.x);
        split_pair_n<PA>(v0.y, v1.y, h.y, m.y, l.y);
        split_pair_n<PA>(v0.z, v1.z, h.z, m.z, l.z);
        split_pair_n<PA>(v0.w, v1.w, h.w, m.w, l.w);
        *reinterpret_cast<uint4*>(d) = h;
        if constexpr (PA == 3) {
            *reinterpret_cast<uint4*>(d + plane_bytes) = m;
            *reinterpret_cast<uint4*>(d + 2 * plane_bytes) = l;
        }
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
            char* dB = dS + PA * SPA;
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
    constexpr bool WIL = SP && TM * TN == 1;      // (the 128 x 128 pair kernels run out of registers: they keep the plain tile,
                                                  //  so both forms are product code and both are under the parity tests)
    auto k_tile = [&](auto buf_c, int kt, auto last_c) __attribute__((always_inline)) {
        constexpr int buf = decltype(buf_c)::value;
        constexpr bool LAST = decltype(last_c)::value;
        if (WIL ? !LAST : kt + 1 < nk) issue_loads(kt + 1);
        // tile kt+1 was the last reader of group (kt+1)/T when it is that group's last tile; the
        // slot is rewritten (group + 2) one iteration later, after this iteration's barrier.
        if (kt % TILES_PER_GROUP == 0 && kt >= TILES_PER_GROUP) produce_rows(((kt / TILES_PER_GROUP) + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (WIL && !LAST && !U8 && SPLIT != 1) {
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
            const unsigned* cB = reinterpret_cast<const unsigned*>(sS + buf * STAGE + PA * SPA) + (half * 4) * BN + wn * TN * 32 + l31;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                u32x4 fa[TM][3], fb[TN][3];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int pl = 0; pl < PA; ++pl) {
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
                split_products<SPLIT, PA, PB, false, TM, TN>(fa, fb, acc);
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
                constexpr int NMW = TM * TN * (SPLIT == 1 ? 1 : PB == 1 ? 3 : SPLIT) * (BK / 16);   // MFMAs per k-tile and wave
                constexpr int NVW = (RA / 2) * (PA == 1 ? 4 : 44) + (RB / 2) * (U8 ? 12 : PB == 1 ? 4 : 44);   // the split's vector instructions
                constexpr int NWW = (RA / 2) * PA + (RB / 2) * PB;                       // its LDS stores
                constexpr int VPMW = (NVW + NMW - 1) / NMW < 6 ? (NVW + NMW - 1) / NMW : 6;
                constexpr int WEVW = NMW / NWW > 0 ? NMW / NWW : 1;
                __builtin_amdgcn_sched_group_barrier(0x100, (BK / 16) * 4 * (TM * PA + TN * PB), 0);   // every fragment read first
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

}  // namespace arlc
