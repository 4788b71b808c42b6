// Launchers, per-call context, split planning and the instantiation lists dealt out to mfma_conv_p1 .. p7.hip.
// Part of mfma_conv_impl.h.
#pragma once
#include "mfma_generic.h"
#include "mfma_igemm.h"
#include "mfma_wgrad.h"
#include "mfma_pair.h"

namespace arlc {

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
    return g->route == ARL_CONV_ROUTE_SPLIT9 ? 9 : g->route == ARL_CONV_ROUTE_FP32 ? 0 : g->route == ARL_CONV_ROUTE_SPLIT6 ? 6
         : g->route == ARL_CONV_ROUTE_BF16 ? 1 : -1;
}
#define ARL_ROUTE_SCOPE(geom, job)                                                                         \
    ARL_REQUIRE(split_of(geom) >= 0, ARL_E_ARG, "conv route: ARL_CONV_ROUTE_SPLIT9, _FP32, _SPLIT6 or _BF16");    \
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
#define ARL_BY_MODE(X1, X6, X9) do { X9; } while (0)
#else
#define ARL_BY_MODE(X1, X6, X9) do { if (g_split == 1) { X1; } else if (g_split == 6) { X6; } else { X9; } } while (0)
#endif

// the launch of igemm_split_kernel: two LDS stages of (1 or 3) + 3 bf16 planes; a data gradient hosts the pending
// optimiser job like launch_igemm_occ
template <int WGM, int WGN, int TM, int TN, int BK, bool B_KC, bool U8, int MINW>
int launch_igemm_split(const GemmArgs& a, bool multi_tap, bool has_pad, hipStream_t s, int splits = 1) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    const int npl = planes_of(g_split);
    const size_t lds = (size_t)2 * ((U8 ? 1 : npl) * BM + npl * BN) * BK * 2;
    const size_t lds_dir = (size_t)2 * npl * BN * BK * 2;       // direct gathered operand: only the weights live in LDS
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
        ARL_BY_MODE(ARL_SPLIT_PIN(MT, HP, 1, CO), ARL_SPLIT_PIN(MT, HP, 6, CO), ARL_SPLIT_PIN(MT, HP, 9, CO));                           \
    } while (0)
    if constexpr (U8) {
        ARL_SPLIT_MODE(false, false, false);
    } else {
        // (data gradients: k-major weights, or k-contiguous ones on the one-wave-per-row-tile shapes)
        if constexpr (!B_KC || (WGM == 4 && WGN == 1 && TM == 1)) {
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
    const int npl = planes_of(g_split);
    const size_t lds = (size_t)2 * (npl * BM + (U8 ? 1 : npl) * BN) * BK * 2;
    dim3 grid((a.N + BN - 1) / BN, (a.K_out + BM - 1) / BM, splits);
    int rc = 0;
#define ARL_WSPLIT(HP, SPL)                                                                                \
    do {                                                                                                   \
        auto k = wgrad_split_kernel<WGM, WGN, TM, TN, BK, HP, U8, SPL, MINW>;                              \
        rc = allow_big_lds(k, lds + 4096);                                                                 \
        if (!rc) hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);                                        \
    } while (0)
    if constexpr (U8) {
        ARL_BY_MODE(ARL_WSPLIT(false, 1), ARL_WSPLIT(false, 6), ARL_WSPLIT(false, 9));
    } else if (has_pad) {
        ARL_BY_MODE(ARL_WSPLIT(true, 1), ARL_WSPLIT(true, 6), ARL_WSPLIT(true, 9));
    } else {
        ARL_BY_MODE(ARL_WSPLIT(false, 1), ARL_WSPLIT(false, 6), ARL_WSPLIT(false, 9));
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
        const size_t lds_s = (size_t)2 * planes_of(g_split) * ((DBM + DBN) > (WBM + WBN) ? (DBM + DBN) : (WBM + WBN)) * BK * 2;
#define ARL_PSPLIT(HP, SPL)                                                                                \
    do {                                                                                                   \
        auto k = bwd_pair_kernel<DWGM, DWGN, DTM, DTN, WWGM, WWGN, WTM, WTN, BK, HP, SPL>;                 \
        rc = allow_big_lds(k, lds_s + 4096);                                                               \
        if (!rc) hipLaunchKernelGGL(k, dim3(n_ig + n_wg), dim3(256), lds_s, s, d.a, w.a, dgx, dgy, n_ig, wgx, wgy); \
    } while (0)
        if (has_pad) ARL_BY_MODE(ARL_PSPLIT(true, 1), ARL_PSPLIT(true, 6), ARL_PSPLIT(true, 9));
        else ARL_BY_MODE(ARL_PSPLIT(false, 1), ARL_PSPLIT(false, 6), ARL_PSPLIT(false, 9));
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
                     hipStream_t s, unsigned char* copy_out = nullptr, long long copy_stride = 0, int* zero_word = nullptr);

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
