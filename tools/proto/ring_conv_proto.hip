// Prototype (round 5): the 64-column forward convolutions as a PERSISTENT, role-split, ring-fed implicit GEMM.
//
//   * one 512-thread workgroup per CU: waves 0-3 COMPUTE (one per SIMD: a 32-row x 64-column tile each, the nine exact
//     bf16-split products of the production kernels, same product and k order), waves 4-7 LOAD (LDS-DMA only);
//   * an S-slot LDS ring of k-tiles (32 reduction indices): the gathered operand as FULL 128-byte lines
//     (`buffer_load_dwordx4 ... lds`, 8 rows x 128 B per instruction -- 8 cache lines, where the production kernel's
//     fragment-shaped loads touch 32 lines per instruction), 16-byte chunks XOR-swizzled on the SOURCE side so that the
//     compute waves' ds_read_b128 are conflict-free; the weights pre-split ONCE per optimiser step into ready-made LDS
//     images (three bf16 planes per k-tile), copied linearly -- no registers, no vector work, no LDS stores for either
//     operand in the compute waves;
//   * the loaders run S - 1 k-tiles ahead of the MFMAs, across tile boundaries;
//   * the launch's (tile, k-tile) units are dealt out EVENLY to the workgroups (432 tiles x 16 k-tiles = 27 units per CU
//     instead of "two tiles on 176 CUs, one on 80"): a tile cut between two workgroups is finished by the one that holds
//     its k = 0 end, which adds the other's partial sums (published through global memory with write-through stores and a
//     per-wave flag, long before they are needed) -- a FIXED schedule, so results reproduce bit for bit.
// Checks against a float64 contraction and times it.  usage: ring_conv_proto [images] [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <type_traits>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
constexpr unsigned HI16 = 0xffff0000u;
__device__ __forceinline__ unsigned hi_pair(float x0, float x1) {
    return __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
}
__device__ __forceinline__ float lo_part(float x) { return x - __uint_as_float(__float_as_uint(x) & HI16); }
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = hi_pair(x0, x1);
    const float r0 = lo_part(x0), r1 = lo_part(x1);
    m = hi_pair(r0, r1);
    l = hi_pair(lo_part(r0), lo_part(r1));
}

constexpr int BN = 64, BK = 32;
constexpr int W_PLANE = BN * BK * 2;            // 4 KB:  [col][64 B],  16-byte chunk q of col n at slot q ^ ((n >> 2) & 3)
constexpr int W_BYTES = 3 * W_PLANE;
constexpr int a_bytes(int cw) { return 32 * cw * BK * 4; }      // [row][128 B], 16-byte chunk c of row r at slot c ^ ((r >> 1) & 7)
constexpr int slot_bytes(int cw) { return a_bytes(cw) + W_BYTES; }   // 28 KB at 4 compute waves, 44 KB at 8
#ifndef DMA_PER_UNIT_
#define DMA_PER_UNIT_ 7
#endif

struct RingArgs {
    const float* x; float* y; const float* bias; const char* wimg; const float* zero_page;
    float* partial; unsigned* flags; unsigned epoch;
    int n_img, H, W, OH, OW, M, n_tiles, kt_per_tile, n_units;
    unsigned x_bytes;
    unsigned long long* trace;
};

// W[n][K] (K index = (ty * kw + tx) * C + c) -> per k-tile the LDS image: 3 planes x [64 cols][32 k] bf16, swizzled
__global__ void wprep_ring_kernel(const float* w, char* img, int K) {
    const int n_kt = K / BK;
    const int total = n_kt * BN * 4;                    // one thread per (k-tile, col, 8-k chunk)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int q = i & 3, n = (i >> 2) % BN, kt = i / (4 * BN);
        const float* src = w + (size_t)n * K + kt * BK + q * 8;
        unsigned h[4], m[4], l[4];
        for (int p = 0; p < 4; ++p) split_pair(src[2 * p], src[2 * p + 1], h[p], m[p], l[p]);
        char* d = img + (size_t)kt * W_BYTES + n * 64 + ((q ^ ((n >> 2) & 3)) << 4);
        *reinterpret_cast<u32x4*>(d) = u32x4{h[0], h[1], h[2], h[3]};
        *reinterpret_cast<u32x4*>(d + W_PLANE) = u32x4{m[0], m[1], m[2], m[3]};
        *reinterpret_cast<u32x4*>(d + 2 * W_PLANE) = u32x4{l[0], l[1], l[2], l[3]};
    }
}

// unit range of workgroup g of G: [g * U / G, (g + 1) * U / G)
__host__ __device__ inline int unit_begin(int g, int G, int U) { return (int)(((long long)g * U) / G); }

// FLAGS: no workgroup barrier in the main loop.  The ring is a producer / consumer queue on LDS counters: ready[slot]
// counts loader-wave arrivals (4 per unit), done[slot] compute-wave releases (CW per unit); a compute wave checks the
// counter of the NEXT unit half a unit before it reads it (the check's LDS read rides with the fragment reads), a loader
// wave spins (s_sleep) until the slot it is about to refill has been released by every compute wave.  With a barrier per
// unit the waves of a workgroup stay in phase: the two compute waves of a SIMD then reach their pipeline fill / drain
// together and the matrix pipe idles ~800 of every 3 100 cycles (KO "no barrier": 2 300).
template <int C, int KH, int KW, int STRIDE, int PAD, int S, bool BUFDMA, int KO = 0, int CW = 4, bool FLAGS = false>
__global__ __launch_bounds__(64 * (CW + 4)) void ring_conv_kernel(const RingArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    volatile unsigned* const ctrl = reinterpret_cast<volatile unsigned*>(lds + S * slot_bytes(CW));   // ready[S], done[S]
    if (FLAGS) {
        if (threadIdx.x < 2 * S) ctrl[threadIdx.x] = 0;
        __syncthreads();
    }
    constexpr int BM = 32 * CW, A_BYTES = a_bytes(CW), SLOT = slot_bytes(CW), NAP = CW;   // NAP: A pieces per loader wave
    static_assert(CW == 4 || CW == 8, "4 compute waves (one per SIMD) or 8 (two)");
    constexpr int CB = C / BK;                          // k-tiles per filter tap
    constexpr int D = S - 1;                            // k-tiles the loaders run ahead
    constexpr int DMA_PER_UNIT = (KO & 16) ? 3 : (KO & 32) ? CW : CW + 3;   // per loader wave: CW pieces of the gathered operand, 3 of the weights
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int G = gridDim.x, g = blockIdx.x;
    const int u0 = unit_begin(g, G, a.n_units), u1 = unit_begin(g + 1, G, a.n_units);
    const int n = u1 - u0, KT = a.kt_per_tile;
    unsigned long long t0 = 0, t1 = 0;
    if (a.trace) t0 = __builtin_readcyclecounter();

    if (wave >= CW) {
        if (KO & 2048) return;                           // (with KO 1: the compute waves have the CU to themselves)
        // ================================================================ loaders
        const int L = wave - CW;
        const int sub = lane >> 3;                      // row within an 8-row piece
        const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.wimg), 0, (unsigned)(KT * W_BYTES), 0x00020000);
        int cur_tile = -1;
        int rbase[NAP];                                 // element offset of the row's tap origin (may be negative)
        unsigned vmask[NAP];                            // bit ty * KW + tx set = tap inside the image
        auto decode = [&](int tile) {
#pragma unroll
            for (int i = 0; i < NAP; ++i) {
                const int row = 8 * NAP * L + 8 * i + sub, m = tile * BM + row;
                const int b = m / (a.OH * a.OW), r = m - b * (a.OH * a.OW), oy = r / a.OW, ox = r - oy * a.OW;
                const int y0 = oy * STRIDE - PAD, x0 = ox * STRIDE - PAD;
                rbase[i] = ((b * a.H + y0) * a.W + x0) * C;
                unsigned mk = 0;
                for (int ty = 0; ty < KH; ++ty)
                    for (int tx = 0; tx < KW; ++tx)
                        if (m < a.M && y0 + ty >= 0 && y0 + ty < a.H && x0 + tx >= 0 && x0 + tx < a.W) mk |= 1u << (ty * KW + tx);
                vmask[i] = mk;
            }
        };
        auto issue = [&](int j) {                        // unit j of this workgroup -> ring slot j % S
            const int u = u0 + j, tile = u / KT, kt = u - tile * KT;
            if (tile != cur_tile) { decode((KO & 64) ? (tile & 7) : tile); cur_tile = tile; }        // uniform (KO 64: eight L2-hot tiles)
            const int tap = kt / CB, ch0 = (kt - tap * CB) * BK, ty = tap / KW, tx = tap - ty * KW;
            const int tap_off = (ty * a.W + tx) * C + ch0;
            char* slot = lds + (j % S) * SLOT;
            if (KO & 4) return;
#pragma unroll
            for (int i = 0; i < ((KO & 16) ? 0 : NAP); ++i) {
                const int row = 8 * NAP * L + 8 * i + sub;
                const int chunk = (lane & 7) ^ ((row >> 1) & 7);
                const bool ok = (vmask[i] >> tap) & 1;
                char* dst = slot + (8 * NAP * L + 8 * i) * 128;
                if constexpr (BUFDMA) {                  // out-of-range offset: the hardware's bounds check writes zeros
                    const unsigned off = ok ? (unsigned)(rbase[i] + tap_off + chunk * 4) << 2 : 0x7ffffff0u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, LDS_PTR(dst), 16, off, 0, 0, 0);
                } else {
                    const float* src = ok ? a.x + (rbase[i] + tap_off + chunk * 4) : a.zero_page + (lane & 7) * 4;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, LDS_PTR(dst), 16, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < ((KO & 32) ? 0 : 3); ++i) {
                const int piece = L + 4 * i;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(slot + A_BYTES + piece * 1024), 16,
                                                         (unsigned)(kt * W_BYTES + piece * 1024 + lane * 16), 0, 0, 0);
            }
        };
        // The compute waves read their fragments one 16-k step AHEAD of their MFMAs -- during unit `it` they already read
        // the first half of unit it + 1 -- so unit it + 1 must be in the ring when iteration `it` begins, and the slot the
        // loaders refill during iteration `it` is the one of unit it - 1: D = S - 1 units issued ahead, all but the newest
        // one landed at every barrier.
        if constexpr (FLAGS) {
            constexpr int Q = 2;                         // units a loader wave keeps in flight
            for (int j = 0; j < n; ++j) {
                const int slot_j = j % S;
                if (j >= S) {                            // every compute wave has released unit j - S
                    const unsigned need = (unsigned)CW * (unsigned)(j / S);
                    while (__builtin_amdgcn_readfirstlane(ctrl[S + slot_j]) < need) __builtin_amdgcn_s_sleep(1);
                }
                issue(j);
                if (j >= Q - 1) {                        // unit j - (Q - 1) has landed: say so
                    asm volatile("s_waitcnt vmcnt(%0)" :: "i"(DMA_PER_UNIT * (Q - 1)) : "memory");
                    if (lane == 0) __hip_atomic_fetch_add(const_cast<unsigned*>(&ctrl[(j - (Q - 1)) % S]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int j = n - (Q - 1) < 0 ? 0 : n - (Q - 1); j < n; ++j)
                if (lane == 0) __hip_atomic_fetch_add(const_cast<unsigned*>(&ctrl[j % S]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return;
        }
        for (int j = 0; j < D && j < n; ++j) issue(j);
        if (n > D - 1 && D >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(DMA_PER_UNIT * (D - 2)) : "memory");   // units 0, 1 landed
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int it = 0; it < n; ++it) {
            if (it + D < n) {
                issue(it + D);                           // its slot was last read in iteration it - 1
                asm volatile("s_waitcnt vmcnt(%0)" :: "i"(DMA_PER_UNIT * (D - 2)) : "memory");    // unit it + 2 has landed
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // ==================================================================== compute
    const int w = wave;
    const int arow = 32 * w + l31;
    const unsigned a_off = (unsigned)arow * 128u;
    const unsigned a_swz = (unsigned)((arow >> 1) & 7);
    const unsigned w_off = (unsigned)A_BYTES + (unsigned)l31 * 64u;
    const unsigned w_swz = (unsigned)((l31 >> 2) & 3);
    float* const slab = a.partial + (size_t)g * (BM * BN);       // this workgroup's partial tile (register layout)
    unsigned* const my_flag = a.flags + g * 8 + w;
    int pending_flag = -1;                              // iteration after which the published partial's flag goes out
    auto wait_ready = [&](int j) {                      // unit j of this workgroup is in the ring (all four loader waves)
        const unsigned need = 4u * (unsigned)(j / S + 1);
        while (__builtin_amdgcn_readfirstlane(ctrl[j % S]) < need) __builtin_amdgcn_s_sleep(0);
    };
    if (FLAGS) { if (n > 0) wait_ready(0); }
    else if (!(KO & 2048)) __builtin_amdgcn_s_barrier();     // unit 0 is in the ring
    if (a.trace) t1 = __builtin_readcyclecounter();
    // Software pipeline over 16-k steps: while the 18 MFMAs of step t run, the fragments of step t + 1 are read (2 + 6
    // ds_read_b128) and its gathered operand is split (44 vector instructions) -- register sets t & 1.
    u32x4 fa[2][3], fb[2][2][3];
    float4 xr[2];
    auto read_step = [&](const char* slot, auto ks_c, auto set_c) {
        constexpr int ks = decltype(ks_c)::value, st = decltype(set_c)::value;
        const unsigned q = 2 * ks + half;                // 8-k chunk of the lane: fp32 chunks 2q, 2q + 1
        if (KO & 256) {                                  // keep the registers live and opaque, read nothing
            asm volatile("" : "+v"(xr[0].x), "+v"(xr[0].y), "+v"(xr[0].z), "+v"(xr[0].w), "+v"(xr[1].x), "+v"(xr[1].y), "+v"(xr[1].z), "+v"(xr[1].w));
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) asm volatile("" : "+v"(fb[st][j][pl]));
            return;
        }
        xr[0] = *reinterpret_cast<const float4*>(slot + a_off + (((2 * q) ^ a_swz) << 4));
        xr[1] = *reinterpret_cast<const float4*>(slot + a_off + (((2 * q + 1) ^ a_swz) << 4));
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                fb[st][j][pl] = *reinterpret_cast<const u32x4*>(slot + w_off + pl * W_PLANE + j * 32 * 64 + ((q ^ w_swz) << 4));
    };
    auto split_step = [&](auto set_c) {
        constexpr int st = decltype(set_c)::value;
        if (KO & 512) {
            fa[st][0] = u32x4{__float_as_uint(xr[0].x), __float_as_uint(xr[0].y), __float_as_uint(xr[0].z), __float_as_uint(xr[0].w)};
            fa[st][1] = u32x4{__float_as_uint(xr[1].x), __float_as_uint(xr[1].y), __float_as_uint(xr[1].z), __float_as_uint(xr[1].w)};
            fa[st][2] = fa[st][0];
            return;
        }
        unsigned h[4], m[4], l[4];
        split_pair(xr[0].x, xr[0].y, h[0], m[0], l[0]);
        split_pair(xr[0].z, xr[0].w, h[1], m[1], l[1]);
        split_pair(xr[1].x, xr[1].y, h[2], m[2], l[2]);
        split_pair(xr[1].z, xr[1].w, h[3], m[3], l[3]);
        fa[st][0] = u32x4{h[0], h[1], h[2], h[3]};
        fa[st][1] = u32x4{m[0], m[1], m[2], m[3]};
        fa[st][2] = u32x4{l[0], l[1], l[2], l[3]};
    };
    auto interleave = [&]() {                            // 8 LDS reads first, then one MFMA : three vector instructions
        if (KO & (256 | 512 | 1024)) return;
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
        for (int mm = 0; mm < 18; ++mm) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    xr[0] = xr[1] = make_float4(1.f, 2.f, 3.f, 4.f);
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fb[st][j][pl] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    if (n > 0 && !(KO & 2)) { read_step(lds, I0{}, I0{}); split_step(I0{}); }
    int it = 0;
    while (it < n) {                                    // one segment = this workgroup's k-tiles of one tile
        const int u = u0 + it, tile = u / KT, kt0 = u - tile * KT;
        const int seg_units = (KT - kt0) < (n - it) ? (KT - kt0) : (n - it);
        const bool has_head = kt0 == 0, has_tail = kt0 + seg_units == KT;
        // the accumulators live inside the segment: no copies across the epilogue's control flow
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
        auto mfma_step = [&](auto set_c) {               // the nine products, smallest terms first (split_products<9, 3, 3, SWAP>)
            constexpr int st = decltype(set_c)::value;
            if (KO & 1024) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) asm volatile("" :: "v"(fa[st][pl]), "v"(fb[st][0][pl]), "v"(fb[st][1][pl]));
                return;
            }
#pragma unroll
            for (int sm = 4; sm >= 0; --sm)
#pragma unroll
                for (int pa = 0; pa < 3; ++pa) {
                    const int pb = sm - pa;
                    if (pb < 0 || pb >= 3) continue;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[j] = mfma_bf16(fb[st][j][pb], fa[st][pa], acc[j]);
                        // MFMAs keep their program order (the two accumulators take turns); everything else may move
                        __builtin_amdgcn_sched_barrier(0x7F6);
                    }
                }
        };
        // One 16-k step, hand-scheduled: the eight fragment reads of the NEXT step first, then the 18 MFMAs of this step
        // in the production kernels' order (smallest products first, the two accumulators taking turns), and between
        // them -- from the third MFMA on, when the reads have returned -- the next step's split in slices of three
        // vector instructions; a full scheduling fence after every slot keeps exactly this order.
        auto step = [&](const char* rd_slot, auto ks_rd, auto set_rd, auto set_mm) {
            constexpr int sr = decltype(set_rd)::value, sm = decltype(set_mm)::value;
            const bool stamp = (KO & 4096) && a.trace && it == 5 && sr == 1;
            unsigned long long ts[20];
            if (stamp) ts[0] = __builtin_readcyclecounter();
            read_step(rd_slot, ks_rd, set_rd);
            __builtin_amdgcn_sched_barrier(0);
            if (stamp) ts[1] = __builtin_readcyclecounter();
            constexpr int PA[9] = {2, 1, 2, 0, 1, 2, 0, 1, 0}, PB[9] = {2, 2, 1, 2, 1, 0, 1, 0, 0};
            float t0[4], t1[4], r0[4], r1[4];
            unsigned hh[4], mm[4], ll[4];
            const float xs[8] = {xr[0].x, xr[0].y, xr[0].z, xr[0].w, xr[1].x, xr[1].y, xr[1].z, xr[1].w};
#pragma unroll
            for (int i = 0; i < 9; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    // (volatile asm: these keep their order among themselves; the pins below keep the split's slices
                    //  between them -- scheduling fences alone do not, the slices get sunk behind the last MFMA)
                    if (!(KO & 1024)) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(fb[sm][j][PB[i]]), "v"(fa[sm][PA[i]]));
                    const int sl = 2 * i + j - 2;        // slice of the split (16 slices: 4 pairs x 4 stages)
                    if (sl >= 0 && !(KO & 512)) {
                        const int p = sl >> 2, stg = sl & 3;
                        const float x0 = xs[2 * p], x1 = xs[2 * p + 1];
                        if (stg == 0) {
                            t0[p] = __uint_as_float(__float_as_uint(x0) & HI16);
                            t1[p] = __uint_as_float(__float_as_uint(x1) & HI16);
                            hh[p] = hi_pair(x0, x1);
                            asm volatile("" : "+v"(t0[p]), "+v"(t1[p]), "+v"(hh[p]));
                        } else if (stg == 1) {
                            r0[p] = x0 - t0[p];
                            r1[p] = x1 - t1[p];
                            t0[p] = __uint_as_float(__float_as_uint(r0[p]) & HI16);
                            asm volatile("" : "+v"(r0[p]), "+v"(r1[p]), "+v"(t0[p]));
                        } else if (stg == 2) {
                            t1[p] = __uint_as_float(__float_as_uint(r1[p]) & HI16);
                            mm[p] = hi_pair(r0[p], r1[p]);
                            r0[p] = r0[p] - t0[p];
                            asm volatile("" : "+v"(t1[p]), "+v"(mm[p]), "+v"(r0[p]));
                        } else {
                            r1[p] = r1[p] - t1[p];
                            ll[p] = hi_pair(r0[p], r1[p]);
                            asm volatile("" : "+v"(r1[p]), "+v"(ll[p]));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (stamp) ts[2 + 2 * i + j] = __builtin_readcyclecounter();
                }
            if (stamp && g == 0 && w == 0 && lane == 0)
                for (int q = 0; q < 20; ++q) a.trace[12288 + q] = ts[q];
            if (KO & 512) {
                fa[sr][0] = u32x4{__float_as_uint(xs[0]), __float_as_uint(xs[1]), __float_as_uint(xs[2]), __float_as_uint(xs[3])};
                fa[sr][1] = u32x4{__float_as_uint(xs[4]), __float_as_uint(xs[5]), __float_as_uint(xs[6]), __float_as_uint(xs[7])};
                fa[sr][2] = fa[sr][0];
            } else {
                fa[sr][0] = u32x4{hh[0], hh[1], hh[2], hh[3]};
                fa[sr][1] = u32x4{mm[0], mm[1], mm[2], mm[3]};
                fa[sr][2] = u32x4{ll[0], ll[1], ll[2], ll[3]};
            }
        };
        for (int k = 0; k < seg_units; ++k, ++it) {
            const char* slot = lds + (it % S) * SLOT;
            const char* slot_next = lds + ((it + 1) % S) * SLOT;       // (after the last unit: read, never used)
            if (!(KO & 2)) {
                step(slot, I1{}, I1{}, I0{});            // MFMAs of step (it, 0); fragments of step (it, 1) read and split
                if (FLAGS && it + 1 < n) wait_ready(it + 1);
                step(slot_next, I0{}, I0{}, I1{});       // MFMAs of step (it, 1); fragments of step (it + 1, 0)
                // every read of unit `it` has been issued (and the LDS serves a wave's operations in order): release its slot
                if (FLAGS && lane == 0) __hip_atomic_fetch_add(const_cast<unsigned*>(&ctrl[S + it % S]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (pending_flag == it) {                    // the partial's stores were issued two iterations ago
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(my_flag, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pending_flag = -1;
            }
            if (!FLAGS && !(KO & 1)) __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");     // the last MFMA's result (asm: the compiler pads nothing)
        if (!has_head) {
            // a tail piece (the first thing this workgroup computed): publish the partial sums, write-through, in
            // register layout; the flag follows once the stores have drained (two iterations on, or at the end)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    u32x4 raw = {__float_as_uint(acc[j][4 * q]), __float_as_uint(acc[j][4 * q + 1]),
                                 __float_as_uint(acc[j][4 * q + 2]), __float_as_uint(acc[j][4 * q + 3])};
                    u32x4* dst = reinterpret_cast<u32x4*>(slab) + ((w * 8 + j * 4 + q) * 64 + lane);
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(raw) : "memory");
                }
            pending_flag = it + 1 < n ? it + 1 : -2;
        } else {
            if (!has_tail) {
                // a head piece (the LAST thing this workgroup computes): the rest of the tile is workgroup g + 1's first
                // piece -- wait for its flag (set long ago), add its partial sums
                const unsigned* flag = a.flags + (g + 1) * 8 + w;
                if (!(KO & 8)) {
                    unsigned spins = 0;
                    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch && ++spins < (1u << 24))
                        __builtin_amdgcn_s_sleep(2);
                }
                // all eight 16-byte loads in flight together (L2-served: the producer stored write-through)
                const u32x4* sA = reinterpret_cast<const u32x4*>(a.partial + (size_t)(g + 1) * (BM * BN)) + (w * 8 * 64 + lane);
                const u32x4* sB = sA + 4 * 64;
                u32x4 p[8];
                asm volatile("global_load_dwordx4 %0, %8, off sc1\n\t"
                             "global_load_dwordx4 %1, %8, off offset:1024 sc1\n\t"
                             "global_load_dwordx4 %2, %8, off offset:2048 sc1\n\t"
                             "global_load_dwordx4 %3, %8, off offset:3072 sc1\n\t"
                             "global_load_dwordx4 %4, %9, off sc1\n\t"
                             "global_load_dwordx4 %5, %9, off offset:1024 sc1\n\t"
                             "global_load_dwordx4 %6, %9, off offset:2048 sc1\n\t"
                             "global_load_dwordx4 %7, %9, off offset:3072 sc1\n\t"
                             "s_waitcnt vmcnt(0)"
                             : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]), "=&v"(p[7])
                             : "v"(sA), "v"(sB) : "memory");
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const u32x4 v4 = p[j * 4 + q];
                        acc[j][4 * q] += __uint_as_float(v4.x); acc[j][4 * q + 1] += __uint_as_float(v4.y);
                        acc[j][4 * q + 2] += __uint_as_float(v4.z); acc[j][4 * q + 3] += __uint_as_float(v4.w);
                    }
            }
            // epilogue: the lane owns output row m and, per column tile j and quad q, four consecutive channels
            const int m = tile * BM + arow;
            if (m < a.M) {
                float* out = a.y + (size_t)m * BN;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int nn = j * 32 + 8 * q + 4 * half;
                        const float4 b = *reinterpret_cast<const float4*>(a.bias + nn);
                        float4 v = make_float4(acc[j][4 * q] + b.x, acc[j][4 * q + 1] + b.y, acc[j][4 * q + 2] + b.z, acc[j][4 * q + 3] + b.w);
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                        *reinterpret_cast<float4*>(out + nn) = v;
                    }
            }
        }
    }
    if (pending_flag == -2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(my_flag, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (a.trace && lane == 0) {
        unsigned long long* t = a.trace + ((size_t)g * CW + w) * 4;
        t[0] = t0; t[1] = t1; t[2] = __builtin_readcyclecounter(); t[3] = n;
    }
}

__global__ void ref_conv_kernel(const float* x, const float* w, const float* bias, double* y, int n_img, int H, int W, int C,
                                int N, int kh, int kw, int stride, int pad, int OH, int OW) {
    const size_t total = (size_t)n_img * OH * OW * N;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = i % N; size_t r = i / N;
        const int ox = r % OW; r /= OW; const int oy = r % OH; const int b = r / OH;
        double s = 0;
        for (int ty = 0; ty < kh; ++ty)
            for (int tx = 0; tx < kw; ++tx) {
                const int yy = oy * stride - pad + ty, xx = ox * stride - pad + tx;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                const float* xp = x + ((size_t)(b * H + yy) * W + xx) * C;
                const float* wp = w + ((size_t)n * kh * kw + ty * kw + tx) * C;
                for (int c = 0; c < C; ++c) s += (double)xp[c] * (double)wp[c];
            }
        s += bias[n];
        y[i] = s > 0 ? s : 0;
    }
}

template <typename K>
float run(K k, const char* what, RingArgs a, int grid, size_t lds_bytes, int reps, unsigned* epoch, int threads = 512) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) { a.epoch = ++*epoch; hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds_bytes, 0, a); }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) { a.epoch = ++*epoch; hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds_bytes, 0, a); }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // one more launch with per-wave timestamps
    static unsigned long long* dtr = nullptr;
    if (!dtr) CK(hipMalloc(&dtr, (size_t)16384 * 8));
    CK(hipMemset(dtr, 0, 16384 * 8));
    RingArgs at = a; at.trace = dtr; at.epoch = ++*epoch;
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds_bytes, 0, at);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> tr((size_t)grid * 32);
    CK(hipMemcpy(tr.data(), dtr, (size_t)grid * 32 * 8, hipMemcpyDeviceToHost));
    double pro = 0, loop = 0, units = 0; unsigned long long first = ~0ull, last = 0;
    const int cw = threads / 64 - 4;
    for (int i = 0; i < grid * cw; ++i) {
        pro += tr[4 * i + 1] - tr[4 * i]; loop += tr[4 * i + 2] - tr[4 * i + 1]; units += tr[4 * i + 3];
        if (tr[4 * i] < first) first = tr[4 * i];
        if (tr[4 * i + 2] > last) last = tr[4 * i + 2];
    }
    {
        std::vector<unsigned long long> st(20);
        CK(hipMemcpy(st.data(), dtr + 12288, 20 * 8, hipMemcpyDeviceToHost));
        if (st[0]) {
            printf("      step timeline (cycles since the top): reads issued %llu |", st[1] - st[0]);
            for (int q = 2; q < 20; ++q) printf(" %llu", st[q] - st[0]);
            printf("\n");
        }
    }
    printf("%-62s %7.2f us | per compute wave: %6.0f cycles to the first unit, loop %6.0f = %5.0f per k-tile\n", what, ms / reps * 1e3,
           pro / (grid * cw), loop / (grid * cw), loop / units);
    return ms / reps * 1e3f;
}

template <int C, int KH, int KW, int STRIDE, int PAD, int CW>
void layer(const char* name, int n_img, int H, int W, int reps) {
    constexpr int BM = 32 * CW;
    const int N = 64, OH = (H + 2 * PAD - KH) / STRIDE + 1, OW = (W + 2 * PAD - KW) / STRIDE + 1, K = KH * KW * C;
    const size_t nx = (size_t)n_img * H * W * C, ny = (size_t)n_img * OH * OW * N, nw = (size_t)N * K;
    std::vector<float> hx(nx), hw(nw), hb(N);
    srand(7);
    for (auto& v : hx) { float r = (rand() / (float)RAND_MAX) * 2.f - 0.8f; v = r > 0 ? r : 0; }
    for (auto& v : hw) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.2f;
    for (auto& v : hb) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.1f;
    float *dx, *dw, *db, *dy, *dzero, *dpart; double* dref; char* dimg; unsigned* dflags;
    const int cus = 256;
    CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&db, N * 4)); CK(hipMalloc(&dy, ny * 4));
    CK(hipMalloc(&dref, ny * 8)); CK(hipMalloc(&dimg, (size_t)(K / BK) * W_BYTES)); CK(hipMalloc(&dzero, 256));
    CK(hipMalloc(&dpart, (size_t)(cus + 1) * 256 * BN * 4)); CK(hipMalloc(&dflags, (cus + 1) * 8 * 4));
    CK(hipMemset(dzero, 0, 256)); CK(hipMemset(dflags, 0, (cus + 1) * 8 * 4));
    CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(wprep_ring_kernel, dim3(64), dim3(256), 0, 0, dw, dimg, K);
    hipLaunchKernelGGL(ref_conv_kernel, dim3(2048), dim3(256), 0, 0, dx, dw, db, dref, n_img, H, W, C, N, KH, KW, STRIDE, PAD, OH, OW);
    CK(hipDeviceSynchronize());
    RingArgs a = {};
    a.x = dx; a.y = dy; a.bias = db; a.wimg = dimg; a.zero_page = dzero; a.partial = dpart; a.flags = dflags;
    a.n_img = n_img; a.H = H; a.W = W; a.OH = OH; a.OW = OW; a.M = n_img * OH * OW; a.n_tiles = (a.M + BM - 1) / BM;
    a.kt_per_tile = K / BK; a.n_units = a.n_tiles * a.kt_per_tile; a.x_bytes = (unsigned)(nx * 4);
    // every workgroup needs at least one whole tile's worth of units (a tile is then cut at most once)
    int grid = cus;
    while (grid > 1 && a.n_units / grid < a.kt_per_tile) --grid;
    printf("== %s, %d compute waves per workgroup, %d images: M %d (%d tiles), K %d (%d k-tiles per tile), %d units = %.2f per workgroup on %d workgroups; %.2f GF fp32 (x9 bf16)\n",
           name, CW, n_img, a.M, a.n_tiles, K, a.kt_per_tile, a.n_units, a.n_units / (double)grid, grid, 2.0 * a.M * N * K * 1e-9);
    unsigned epoch = 0;
    auto check = [&](const char* tag, float us) {
        std::vector<float> hy(ny); std::vector<double> href(ny);
        CK(hipMemcpy(hy.data(), dy, ny * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(href.data(), dref, ny * 8, hipMemcpyDeviceToHost));
        double worst = 0, big = 0; size_t bad = 0;
        for (size_t i = 0; i < ny; ++i) {
            const double d = fabs((double)hy[i] - href[i]);
            if (!(d <= worst)) worst = d;
            if (fabs(href[i]) > big) big = fabs(href[i]);
            if (!(d <= 1e-5 * (1 + fabs(href[i])))) ++bad;
        }
        printf("   %s: max |err| %.3g (max |ref| %.3g), %zu of %zu outside 1e-5: %s; %.0f TF/s fp32-equivalent, %.0f TF/s on the bf16 pipe\n",
               tag, worst, big, bad, ny, bad ? "FAIL" : "ok", 2.0 * a.M * N * K / us * 1e-6, 18.0 * a.M * N * K / us * 1e-6);
    };
    float us;
    const int threads = 64 * (CW + 4);
    constexpr int SLOT = slot_bytes(CW);
#define VAR(S_, BUF_, KO_) ring_conv_kernel<C, KH, KW, STRIDE, PAD, S_, BUF_, KO_, CW>
#define VARF(S_, KO_) ring_conv_kernel<C, KH, KW, STRIDE, PAD, S_, true, KO_, CW, true>
    if (CW == 4) {
        CK(hipMemset(dy, 0xff, ny * 4));
        us = run(VAR(4, false, 0), "ring 4 slots, global_load_lds + zero page", a, grid, 4 * SLOT, reps, &epoch, threads);
        check("S=4 global", us);
        CK(hipMemset(dy, 0xff, ny * 4));
        us = run(VAR(4, true, 0), "ring 4 slots, buffer_load lds (bounds check = padding)", a, grid, 4 * SLOT, reps, &epoch, threads);
        check("S=4 buffer", us);
    }
    CK(hipMemset(dy, 0xff, ny * 4));
    us = run(VAR(3, true, 0), "ring 3 slots, buffer_load lds", a, grid, 3 * SLOT, reps, &epoch, threads);
    check("S=3 buffer", us);
    CK(hipMemset(dy, 0xff, ny * 4));
    us = run(VARF(3, 0), "ring 3 slots, LDS counters instead of barriers", a, grid, 3 * SLOT + 64, reps, &epoch, threads);
    check("S=3 flags", us);
    if (CW == 4) {
        CK(hipMemset(dy, 0xff, ny * 4));
        us = run(VARF(4, 0), "ring 4 slots, LDS counters instead of barriers", a, grid, 4 * SLOT + 64, reps, &epoch, threads);
        check("S=4 flags", us);
        CK(hipMemset(dy, 0xff, ny * 4));
        us = run(VARF(5, 0), "ring 5 slots, LDS counters instead of barriers", a, grid, 5 * SLOT + 64, reps, &epoch, threads);
        check("S=5 flags", us);
    }
    run(VAR(3, true, 1), "   KO: no barrier (results wrong)", a, grid, 3 * SLOT, reps, &epoch, threads);
    run(VAR(3, true, 2), "   KO: no LDS reads / split / MFMAs (loaders alone)", a, grid, 3 * SLOT, reps, &epoch, threads);
    run(VAR(3, true, 4), "   KO: no DMA (compute alone, results wrong)", a, grid, 3 * SLOT, reps, &epoch, threads);
    run(VAR(3, true, 4 + 1 + 2048), "   KO: compute alone, no barrier, loader waves exit at once", a, grid, 3 * SLOT, reps, &epoch, threads);
    run(VAR(3, true, 4 + 256 + 512 + 1 + 2048), "   KO: MFMAs only, no barrier, loader waves exit at once", a, grid, 3 * SLOT, reps, &epoch, threads);
    run(VAR(3, true, 4 + 1024), "   KO: no DMA, LDS reads + split (no MFMAs)", a, grid, 3 * SLOT, reps, &epoch, threads);
#undef VAR
#undef VARF
    CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dy)); CK(hipFree(dref)); CK(hipFree(dimg)); CK(hipFree(dzero));
    CK(hipFree(dpart)); CK(hipFree(dflags));
}

int main(int argc, char** argv) {
    const int n_img = argc > 1 ? atoi(argv[1]) : 512;
    const int reps = argc > 2 ? atoi(argv[2]) : 50;
    layer<32, 4, 4, 2, 1, 4>("conv2 fwd (25x19x32 -> 12x9x64, 4x4 s2 p1)", n_img, 25, 19, reps);
    layer<32, 4, 4, 2, 1, 8>("conv2 fwd (25x19x32 -> 12x9x64, 4x4 s2 p1)", n_img, 25, 19, reps);
    layer<64, 3, 3, 1, 1, 4>("conv3 fwd (12x9x64 -> 12x9x64, 3x3 s1 p1)", n_img, 12, 9, reps);
    layer<64, 3, 3, 1, 1, 8>("conv3 fwd (12x9x64 -> 12x9x64, 3x3 s1 p1)", n_img, 12, 9, reps);
    return 0;
}
