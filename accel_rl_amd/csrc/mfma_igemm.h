// Scalar-addressed implicit GEMM: conv / dense forward and data gradient (igemm_body and its kernels: fp32 MFMA chain,
// 16-wide tiles, u8 input, bf16-split products with the direct gathered operand).  Part of mfma_conv_impl.h.
#pragma once
#include "mfma_common.h"

namespace arlc {

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
    constexpr int PA = U8 ? 1 : planes_of(SPLIT), PB = planes_of(SPLIT), ROWB = BK * 2, NS = BK / 8;
    constexpr int LPA = ADIR ? 0 : PA;              // planes of the gathered operand that live in LDS
    constexpr int SPA = BM * ROWB, SPB = BN * ROWB, STAGE = LPA * SPA + PB * SPB;
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
    // (a data gradient on k-contiguous weights: this parity class's matrix)
    const unsigned w_cls = (B_KC && a.n_par) ? (unsigned)bz * (unsigned)a.b.cls : 0u;
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(a.b.w + w_cls, a.b.w_bytes - 4u * w_cls);

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
    // (the direct operand two tiles ahead -- loaded a whole tile before its split -- was measured in round 5 and was
    //  slower: conv 2 / conv 3 forward 34.0 / 35.9 -> 36.9 / 39.1 us; profiles/r05/ahead2_ab.txt, LABNOTES.md)
    float4 rd_[TM][DST][2];
    unsigned rd8_[TM][DST][2];
    u32x4 fd_[2][TM][DST][3];
    auto issue_A = [&](auto rs_c) {
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
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const unsigned off = HAS_PAD ? mask_off(imaskD[i], bit, voffD[i]) : voffD[i];
#pragma unroll
            for (int ks = 0; ks < DST; ++ks) {
                rd_[i][ks][0] = buf_ld4s(rsA, off + ks * 64, soffA);
                rd_[i][ks][1] = buf_ld4s(rsA, off + ks * 64 + 16, soffA);
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
                    const float4 q0 = rd_[i][ks][0], q1 = rd_[i][ks][1];
                    unsigned h[4], m[4], l[4];
                    split_pair_n<PA>(q0.x, q0.y, h[0], m[0], l[0]);
                    split_pair_n<PA>(q0.z, q0.w, h[1], m[1], l[1]);
                    split_pair_n<PA>(q1.x, q1.y, h[2], m[2], l[2]);
                    split_pair_n<PA>(q1.z, q1.w, h[3], m[3], l[3]);
                    fd_[rs][i][ks][0] = u32x4{h[0], h[1], h[2], h[3]};
                    if constexpr (PA == 3) {
                        fd_[rs][i][ks][1] = u32x4{m[0], m[1], m[2], m[3]};
                        fd_[rs][i][ks][2] = u32x4{l[0], l[1], l[2], l[3]};
                    }
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
                    split_pair_n<PA>(va[p].x, va[p].y, h.x, m.x, l.x);
                    split_pair_n<PA>(va[p].z, va[p].w, h.y, m.y, l.y);
                    *reinterpret_cast<uint2*>(d) = h;
                    if constexpr (PA == 3) {
                        *reinterpret_cast<uint2*>(d + SPA) = m;
                        *reinterpret_cast<uint2*>(d + 2 * SPA) = l;
                    }
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
                    split_pair_n<PB>(vb[p].x, vb[p].y, h.x, m.x, l.x);
                    split_pair_n<PB>(vb[p].z, vb[p].w, h.y, m.y, l.y);
                    *reinterpret_cast<uint2*>(d) = h;
                    if constexpr (PB == 3) {
                        *reinterpret_cast<uint2*>(d + SPB) = m;
                        *reinterpret_cast<uint2*>(d + 2 * SPB) = l;
                    }
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
                    split_pair_n<PB>(v0.x, v1.x, h.x, m.x, l.x);
                    split_pair_n<PB>(v0.y, v1.y, h.y, m.y, l.y);
                    split_pair_n<PB>(v0.z, v1.z, h.z, m.z, l.z);
                    split_pair_n<PB>(v0.w, v1.w, h.w, m.w, l.w);
                    *reinterpret_cast<uint4*>(d) = h;
                    if constexpr (PB == 3) {
                        *reinterpret_cast<uint4*>(d + SPB) = m;
                        *reinterpret_cast<uint4*>(d + 2 * SPB) = l;
                    }
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
                    for (int pl = 0; pl < PB; ++pl) {
                        if constexpr (B_KC) {
                            fb[j][pl] = *reinterpret_cast<const u32x4*>(cB + pl * SPB + j * 32 * ROWB + slot);
                        } else {
                            const unsigned* q = reinterpret_cast<const unsigned*>(cB + pl * SPB + (ks * 8 * BN + j * 32) * 4);
                            fb[j][pl] = u32x4{q[0], q[BN], q[2 * BN], q[3 * BN]};
                        }
                    }
                split_products<SPLIT, PA, PB, true, TM, TN>(fa, fb, acc);
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
            ARL_STAMP(st_a);
            store_tiles(1, 1);
            if constexpr (ADIR) split_A(C1{});
            ARL_STAMP(st_b);
        } else {
            if constexpr (ADIR) issue_A(C0{});
            issue_loads(kbeg, 0);
            next_tile(); issue_loads(kbeg + BK, 1);
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
        constexpr int NPROD = SPLIT == 1 ? 1 : PA == 1 ? 3 : SPLIT;
        constexpr int NM = TM * TN * NPROD * STEPS;                                     // MFMAs per k-tile and wave
        constexpr int NV = SPLIT == 1 ? RA * (U8 ? 6 : 2) + (B_KC ? RB * 2 : (RB / 2) * 4)
                                      : RA * (U8 ? 6 : 22) + (B_KC ? RB * 22 : (RB / 2) * 44);   // the split's vector instructions
        constexpr int NW = RA * PA + (B_KC ? RB * PB : (RB / 2) * PB);                  // its LDS stores
        constexpr int VPM = (NV + NM - 1) / NM < 6 ? (NV + NM - 1) / NM : 6;
        constexpr int WEV = NM / NW > 0 ? NM / NW : 1;
        auto mid_tile = [&](auto buf_c, int kt) {       // tiles 0 .. nk - 2
            constexpr int buf = decltype(buf_c)::value;
            if (kt + 2 < nk) {                          // uniform: tile kt + 2 -> the set tile kt has left
                next_tile();
                issue_loads(kbeg + (kt + 2) * BK, buf);
            }
            if constexpr (ADIR) {                       // tile kt + 1 of the direct operand
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
            __builtin_amdgcn_sched_group_barrier(0x100, STEPS * (TM * PA + TN * PB * (B_KC ? 1 : 4)), 0);
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
    pin_gemm_args(a);
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

}  // namespace arlc
