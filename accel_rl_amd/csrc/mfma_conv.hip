// Entry points of the MFMA contraction kernels (kernels and launchers: mfma_conv_impl.h; their instantiations are
// built in mfma_conv_p1 .. p7.hip so that the translation units compile side by side).
#include "mfma_conv_impl.h"
#include "dgrad_wt_dev.h"

using namespace arlc;

namespace arlc {
thread_local CallCtx t_ctx = {9, nullptr, false};
unsigned long long* g_trace = nullptr;
bool g_force_generic = false;
int g_fwd_tile = -1;            // arl_dev_fwd_tile: -1 = chosen by the launch's size (fwd_impl)
}  // namespace arlc

namespace {
// out[i] = act(sum_z part[z][i] + bias[i % n_bias]), float4 lanes, fixed summation order: ZG threads share one output
// float4 (thread zg sums splits zg, zg + ZG, ...), then the ZG partial sums are added in index order -- enough parallelism
// for the small, many-split weight gradients without giving up determinism.  ZG = 16, or 64 from FOLD_WIDE splits on
// (round 6: conv 1's weight gradient, 2048 float4 x 512 splits = 16.8 MB, was folded by 128 workgroups whose threads each
// walked 32 dependent-latency loads: 11 us at 1.5 TB/s).  Blocks are 1024 threads: one group of 64 x 16 threads (16
// outputs) in the wide mode, four independent groups of 16 x 16 threads (64 outputs) otherwise -- the arithmetic of an
// item below FOLD_WIDE splits is exactly what the 256-thread kernel of rounds 1-5 did.
constexpr int FOLD_WIDE = 128, FOLD_THREADS = 1024;
int g_fold_wide = FOLD_WIDE;            // arl_dev_fold_wide_from (A/B of the threshold)
bool g_dgrad_wt = true;                 // arl_dev_dgrad_wt: 0 = data gradients ignore the k-contiguous weights (A/B, parity tests)

struct FoldSlot { int64_t i; int zg, zgn, row0, o; };
__device__ __forceinline__ FoldSlot fold_slot(int splits, int local_block, int wide_from) {
    FoldSlot f;
    const int tid = threadIdx.x;
    f.o = tid & 15;
    if (splits >= wide_from) { f.zg = tid >> 4; f.zgn = 64; f.row0 = 0; f.i = (int64_t)local_block * 16 + f.o; }
    else {
        const int grp = tid >> 8;
        f.zg = (tid >> 4) & 15; f.zgn = 16; f.row0 = grp * 16; f.i = ((int64_t)local_block * 4 + grp) * 16 + f.o;
    }
    return f;
}
inline int fold_blocks(int splits, int64_t total4) {
    const int per = splits >= g_fold_wide ? 16 : 64;
    return (int)((total4 + per - 1) / per);
}
__device__ __forceinline__ float4 fold_sum(const float4* __restrict__ part, int splits, int64_t total4, const FoldSlot& f,
                                           float4 (*lds)[16]) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f.i < total4)
        for (int z = f.zg; z < splits; z += f.zgn) {
            const float4 v = part[(int64_t)z * total4 + f.i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    lds[f.row0 + f.zg][f.o] = s;
    __syncthreads();
    if (f.zg == 0 && f.i < total4)
        for (int k = 1; k < f.zgn; ++k) {
            const float4 v = lds[f.row0 + k][f.o];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    return s;
}

__global__ __launch_bounds__(FOLD_THREADS) void fold_splits_kernel(const float4* __restrict__ part, int splits,
                                                                   int64_t total4, const float4* __restrict__ bias,
                                                                   int bias4, int relu, const float4* __restrict__ mask,
                                                                   float4* __restrict__ out, int wide_from) {
    __shared__ float4 lds[64][16];
    const FoldSlot f = fold_slot(splits, (int)blockIdx.x, wide_from);
    float4 s = fold_sum(part, splits, total4, f, lds);
    const int64_t i = f.i;
    if (f.zg == 0 && i < total4) {
        if (bias) {
            const float4 b = bias[i % bias4];
            s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
        }
        if (relu) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
        if (mask) {                                                  // relu backward: 0 where mask <= 0
            const float4 m = mask[i];
            if (!(m.x > 0.f)) s.x = 0.f;
            if (!(m.y > 0.f)) s.y = 0.f;
            if (!(m.z > 0.f)) s.z = 0.f;
            if (!(m.w > 0.f)) s.w = 0.f;
        }
        out[i] = s;
    }
}

// Several independent folds in one launch (arl_fold_many): block -> (item, its outputs) through a block-offset table in
// the kernel arguments; per item the same fixed summation order as above.
struct FoldManyArgs {
    arl_fold_item items[ARL_FOLD_MAX_ITEMS];
    int block_start[ARL_FOLD_MAX_ITEMS + 1];
    int n, wide_from;
};

__global__ __launch_bounds__(FOLD_THREADS) void fold_many_kernel(const FoldManyArgs a) {
    __shared__ float4 lds[64][16];
    int it = 0;
    while (it + 1 < a.n && (int)blockIdx.x >= a.block_start[it + 1]) ++it;      // uniform
    const float4* part = reinterpret_cast<const float4*>(a.items[it].part);
    float4* out = reinterpret_cast<float4*>(a.items[it].out);
    const int64_t total4 = a.items[it].total >> 2;
    const int splits = a.items[it].splits;
    const FoldSlot f = fold_slot(splits, (int)blockIdx.x - a.block_start[it], a.wide_from);
    const float4 s = fold_sum(part, splits, total4, f, lds);
    const int64_t i = f.i;
    if (f.zg == 0 && i < total4) {
        const int64_t valid = a.items[it].valid;            // > 0: `out` holds only this many floats
        if (valid <= 0 || 4 * i + 3 < valid) {
            out[i] = s;
        } else {
            float* o1 = reinterpret_cast<float*>(out + i);
            if (4 * i < valid) o1[0] = s.x;
            if (4 * i + 1 < valid) o1[1] = s.y;
            if (4 * i + 2 < valid) o1[2] = s.z;
        }
    }
}

int launch_fold(const float* part, int splits, int64_t total, const float* bias, int n_bias, int relu,
                float* out, hipStream_t s, const float* mask = nullptr) {
    const int64_t total4 = total >> 2;
    hipLaunchKernelGGL(fold_splits_kernel, dim3((unsigned)fold_blocks(splits, total4)), dim3(FOLD_THREADS), 0, s,
                       (const float4*)part, splits, total4, (const float4*)bias, n_bias >> 2, relu,
                       (const float4*)mask, (float4*)out, g_fold_wide);
    return arl::check_launch("fold_splits_kernel");
}

}  // namespace

extern "C" int64_t arl_conv_workspace_bytes(void) { return (int64_t)64 << 20; }

extern "C" void arl_dev_conv_trace_buffer(void* device_u64_or_null) { g_trace = (unsigned long long*)device_u64_or_null; }

extern "C" void arl_dev_conv_force_generic(int32_t on) { g_force_generic = on != 0; }

extern "C" void arl_dev_fold_wide_from(int32_t splits) { g_fold_wide = splits > 0 ? splits : FOLD_WIDE; }

extern "C" void arl_dev_fwd_tile(int32_t v) { g_fwd_tile = ((v >= 0 && v <= 2) || v == 6 || v == 7) ? v : -1; }

extern "C" void arl_dev_dgrad_wt(int32_t on) { g_dgrad_wt = on != 0; }

extern "C" int arl_corun_job_init(arl_corun_job* job, const arl_opt_state* opt, int32_t method, float learning_rate,
                                  float avg_factor, float beta1_or_rho, float beta2, float epsilon, int32_t k,
                                  float* step_pp, double* norm_parts, int64_t hole_first, int64_t hole_count) {
    ARL_REQUIRE(job, ARL_E_ARG, "null pointer");
    CorunJob* j = reinterpret_cast<CorunJob*>(job);
    int rc = arl::make_opt_seg(&j->seg, opt, method, learning_rate, avg_factor, beta1_or_rho, beta2, epsilon, k,
                               step_pp, norm_parts, hole_first, hole_count, 1, &j->blocks);
    if (rc) return rc;
    j->host_blocks = 256;               // workgroups that run the job inside its host launch (tuning: ARL_CORUN_BLOCKS)
    if (const char* e = getenv("ARL_CORUN_BLOCKS")) j->host_blocks = atoi(e) > 0 ? atoi(e) : 256;
    return 0;
}

extern "C" int arl_corun_job_run(const arl_corun_job* job, void* stream) {
    ARL_REQUIRE(job, ARL_E_ARG, "null pointer");
    const CorunJob* j = reinterpret_cast<const CorunJob*>(job);
    return arl::launch_opt_seg(j->seg, j->blocks, (hipStream_t)stream);
}

namespace {
// parts: leave a split reduction unfolded and describe it (arl_conv2d_fwd_parts); null = fold here
int fwd_impl(const float* x, const float* w, const float* bias_or_null, float* y, const arl_conv_geom* geom, int32_t relu,
             void* workspace, void* stream, arl_fold_item* parts = nullptr) {
    Geom g;
    int rc = check_geom(geom, &g);
    if (rc) return rc;
    ARL_REQUIRE(x && w && y && workspace, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(arl::aligned16(x) && arl::aligned16(w) && arl::aligned16(y) && arl::aligned16(workspace) &&
                    (!bias_or_null || arl::aligned16(bias_or_null)), ARL_E_ALIGN, "16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    GemmArgs a = {};
    a.g.src = x; a.g.Hs = g.H; a.g.Ws = g.W; a.g.Cs = g.C; a.g.out_h = g.Ho; a.g.out_w = g.Wo;
    a.g.mul = g.stride; a.g.add_y = -g.pad_h; a.g.add_x = -g.pad_w; a.g.taps_x = g.kw; a.g.step = 1;
    a.M = (int)(g.batch * g.Ho * g.Wo); a.N = g.K; a.K = g.kh * g.kw * g.C;
    a.g.src_bytes = (unsigned)(g.batch * g.H * g.W * g.C * 4);
    a.b.w = w; a.b.ld = a.K; a.b.w_bytes = (unsigned)((int64_t)a.N * a.K * 4);
    a.o.dense = 1; a.trace = g_trace; a.xcd = 1;
    int splits = 1, per = round_up(a.K, BKT);
    const bool small = a.N >= 128 && (int64_t)a.M * a.N <= (int64_t)1 << 20;     // dense layers: split K
    // (the bf16-split kernels have no 16-wide tiles: 64x64 there)
    const bool small32 = small && !g_split;         // 32x64 tiles: half the splits (and partial bytes) for the same grid
    if (small) plan_split(((a.M + (small32 ? 31 : 63)) / (small32 ? 32 : 64)) * ((a.N + 63) / 64), a.K, &splits, &per, 3 * TARGET_WGS);
    // ... and wider ones whose 128x128 tiles still leave CUs idle (spec-0 dense at the A2C batch: 5120 x 256 =
    // 80 tiles walking 88 k-tiles each, 231 us)
    const int tiles128 = ((a.M + 127) / 128) * ((a.N + 127) / 128);
    if (!small && a.N >= 128 && tiles128 * 4 <= TARGET_WGS * 3) plan_split(tiles128, a.K, &splits, &per, TARGET_WGS);
    // ... and unsplit ones whose 128x128 tiles fill the last round of CUs badly (160 tiles on 256 CUs: 62 %): 64x64 tiles,
    // four times the workgroups, when their fill beats it by more than their lower per-tile efficiency (0.85) -- spec-1
    // dense at 5 120 rows 378.7 -> 309.4 us, 10 240 x 3 456 -> 256: 194.2 -> 160.7, 20 000 x 256 -> 128: 21.1 -> 17.3; 256
    // tiles stay (419 against 459); same k order, bit-identical (profiles/r06/dense_fwd_tile_probe.txt)
    auto cu_fill = [](int t) { return (double)t / (double)(((t + TARGET_WGS - 1) / TARGET_WGS) * TARGET_WGS); };
    const int tiles64 = ((a.M + 63) / 64) * ((a.N + 63) / 64);
    const bool fill64 = !small && a.N >= 128 && splits == 1 && g_split && g_fwd_tile != 7 &&
                        (g_fwd_tile == 6 || 0.85 * cu_fill(tiles64) > cu_fill(tiles128));
    a.k_per_split = per;
    if (splits > 1) {
        ARL_REQUIRE((int64_t)splits * a.M * a.N * 4 <= arl_conv_workspace_bytes(), ARL_E_RANGE, "workspace too small");
        a.o.out = (float*)workspace; a.split_stride = (int64_t)a.M * a.N;
    } else {
        a.o.out = y; a.o.bias = bias_or_null; a.o.relu = relu;
    }
    // scalar-addressed fast path (k-tile of 32 inside one filter row)
    constexpr int FBK = 32;
    const bool single_tap = g.C % FBK == 0;
    const bool multi_tap = !single_tap && FBK % g.C == 0 && g.kw % (FBK / g.C) == 0;
    const bool has_pad = g.pad_h > 0 || g.pad_w > 0;
    const bool fast = !g_force_generic && a.K % FBK == 0 && per % FBK == 0 && (single_tap || multi_tap) &&
                      (!has_pad || g.kh * g.kw <= 32) && a.N % 4 == 0;
    if (fast) {
        a.o.out_bytes = (unsigned)((int64_t)a.M * a.N * 4);     // one split's output
        a.g.mg_w = div_magic((int64_t)a.M + 256, a.g.out_w); a.g.mg_h = div_magic((int64_t)a.M + 256, a.g.out_h);
        a.g.taps_y = g.kh; a.g.dmin = 0;
        a.g.rmin = (a.g.add_y * g.W + a.g.add_x) * g.C;
        a.g.origin = a.g.rmin + a.g.dmin;
        a.g.src_bytes = (unsigned)(g.batch * g.H * g.W * g.C * 4 - (int64_t)a.g.origin * 4);
        // Small tiles keep the LDS footprint low enough for 3-4 workgroups per CU: with one wave per
        // SIMD the barrier, load and LDS latencies of each k-tile sit exposed between MFMA bursts
        // (64x64 instead of 128x64 tiles at N = 64: 46.8 -> 45.0 us, 50.6 -> 48.2 us at the PPO minibatch)
        // (the 128x32 tile at BK = 32 would allow only 3: a 16-wide k-tile, 5-6 resident, is 3-5 us faster)
        if (a.N <= 16 && a.K % 16 == 0 && per % 16 == 0 && (g.C % 16 == 0 || (16 % g.C == 0 && g.kw % (16 / g.C) == 0)))
            rc = launch_igemm<4, 1, 2, 1, 16, true, true>(a, splits, multi_tap, has_pad, s);        // 16-wide MFMA tiles
        else if (g_split && a.N <= 32) rc = launch_igemm_split<4, 1, 1, 1, FBK, true, false, 2>(a, multi_tap, has_pad, s, splits);
        else if (g_split && a.N <= 64) {
            // 33 .. 64 columns.  0: 128 x 64 tiles (a wave owns a 32-row tile and both 32-column halves); 1: 128 x 32
            // tiles, two column tiles per row tile (twice the workgroups, each wave half the MFMA chain; the gathered
            // operand is loaded and split once per column tile); 2: 64 x 64 tiles, both operands through LDS.  All
            // three issue the same piece products in the same k order: bit-identical outputs.
            // By size (round 6, profiles/r06/fwd_tile_probe.txt): while every half tile still gets a CU of its own (at most
            // 128 row tiles: the DQN updates' 32 .. 64-row batches, evaluation groups) the column split is the faster
            // launch (108 row tiles: 19.1 -> 15.6 us); from 129 row tiles on two half workgroups per CU cost what one
            // whole one does, plus the second load and split of the gathered operand (216 row tiles: 19.8 -> 22.7 us).
            const int v = (g_fwd_tile >= 0 && g_fwd_tile <= 2) ? g_fwd_tile : (2 * ((a.M + 127) / 128) <= TARGET_WGS ? 1 : 0);
            if (v == 1) rc = launch_igemm_split<4, 1, 1, 1, FBK, true, false, 2>(a, multi_tap, has_pad, s, splits);
            else if (v == 2) rc = launch_igemm_split<2, 2, 1, 1, FBK, true, false, 3>(a, multi_tap, has_pad, s, splits);
            else rc = launch_igemm_split<4, 1, 1, 2, FBK, true, false, 2>(a, multi_tap, has_pad, s, splits);
        }
        else if (g_split && (small || fill64)) rc = launch_igemm_split<2, 2, 1, 1, FBK, true, false, 3>(a, multi_tap, has_pad, s, splits);
        else if (g_split) rc = launch_igemm_split<2, 2, 2, 2, FBK, true, false, 1>(a, multi_tap, has_pad, s, splits);
        else if (a.N <= 32 && a.K % 16 == 0 && per % 16 == 0 && (g.C % 16 == 0 || (16 % g.C == 0 && g.kw % (16 / g.C) == 0)))
            rc = launch_igemm<4, 1, 1, 1, 16, true>(a, splits, multi_tap, has_pad, s);
        else if (a.N <= 32) rc = launch_igemm<4, 1, 1, 1, FBK, true>(a, splits, multi_tap, has_pad, s);
        else if (a.N <= 64 && splits == 1) rc = launch_n64<true>(a, multi_tap, has_pad, s);
        else if (a.N <= 64) rc = launch_igemm<2, 2, 1, 1, FBK, true>(a, splits, multi_tap, has_pad, s);
        else if (small32) rc = launch_igemm_occ<1, 4, 2, 1, 32, true, true, 5>(a, multi_tap, has_pad, s, splits);
        else if (small) rc = launch_igemm<2, 2, 1, 1, FBK, true>(a, splits, multi_tap, has_pad, s);
        else rc = launch_igemm<2, 2, 2, 2, FBK, true>(a, splits, multi_tap, has_pad, s);
    } else {
        if (a.N <= 32) rc = launch_rowgather<4, 1, 2, 1, 16, true>(a, splits, s);
        else if (a.N <= 64) rc = launch_rowgather<2, 2, 2, 1, 16, true>(a, splits, s);
        else if (small) rc = launch_rowgather<2, 2, 1, 1, 16, true>(a, splits, s);
        else rc = launch_rowgather<2, 2, 2, 2, 16, true>(a, splits, s);
    }
    if (parts) {
        parts->part = splits == 1 ? y : (const float*)workspace; parts->out = y; parts->total = (int64_t)a.M * a.N;
        parts->splits = splits == 1 ? 0 : splits; parts->valid = 0;
        return rc;
    }
    if (rc || splits == 1) return rc;
    return launch_fold((const float*)workspace, splits, (int64_t)a.M * a.N, bias_or_null, a.N, relu, y, s);
}
}  // namespace

namespace arlc {
int fold_wide_from() { return g_fold_wide; }
}

extern "C" int arl_conv2d_fwd_parts(const float* x, const float* w, const float* bias_or_null, float* y,
                                    const arl_conv_geom* geom, int32_t relu, void* workspace, arl_fold_item* item,
                                    void* stream) {
    ARL_REQUIRE(item, ARL_E_ARG, "null pointer");
    ARL_ROUTE_SCOPE(geom, nullptr);
    return fwd_impl(x, w, bias_or_null, y, geom, relu, workspace, stream, item);
}

extern "C" int arl_conv2d_fwd(const float* x, const float* w, const float* bias_or_null, float* y,
                              const arl_conv_geom* geom, int32_t relu, void* workspace, void* stream) {
    ARL_ROUTE_SCOPE(geom, nullptr);
    return fwd_impl(x, w, bias_or_null, y, geom, relu, workspace, stream);
}

namespace {
// plan_only: describe the fast launch instead of issuing it (fast == false: nothing was done)
// workspace (optional): lets a dense layer whose output tiles cannot fill the chip split its reduction
// wt (optional): the same weights as arl_conv2d_dgrad_weights lays them out -- per input-pixel parity class a matrix
// [in_c][taps * out_c] with the reduction index contiguous.  The one-wave-per-row-tile split kernels then read them like a
// forward pass reads its weights (one ds_read_b128 per fragment instead of four ds_read_b32, half the loader's packing
// work): same piece products in the same k order, bit-identical results.
int dgrad_impl(const float* dy, const float* w, const float* mask_or_null, float* dx, const arl_conv_geom* geom,
               DgradPlan* plan_only, void* stream, void* workspace = nullptr, int64_t workspace_bytes = 0,
               const float* wt = nullptr) {
    Geom g;
    int rc = check_geom(geom, &g);
    if (rc) return rc;
    ARL_REQUIRE(dy && w && dx, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(g.kh % g.stride == 0 && g.kw % g.stride == 0, ARL_E_RANGE,
                "data gradient needs kernel size divisible by stride");
    ARL_REQUIRE(arl::aligned16(dy) && arl::aligned16(w) && arl::aligned16(dx) &&
                    (!mask_or_null || arl::aligned16(mask_or_null)) && (!wt || arl::aligned16(wt)), ARL_E_ALIGN,
                "16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    const int st = g.stride;
    constexpr int FBK = 32;
    const int taps_y = g.kh / st, taps_x = g.kw / st;
    const bool has_pad = !(g.kh == 1 && g.kw == 1 && g.pad_h == 0 && g.pad_w == 0);
    const bool fast = !g_force_generic && g.K % FBK == 0 && taps_y * taps_x <= 32 && st * st <= 4 && g.C % 4 == 0;
    // One implicit GEMM per input-pixel parity class (ph, pw): pixels (h, w) = (st*oy + ph, st*ox + pw)
    // only see the taps i = i0 + st*ti, j = j0 + st*tj, which reach output row
    // (h + pad - i) / st = oy + (ph + pad - i0)/st - ti.
    auto describe = [&](int ph, int pw) {
        const int i0 = (ph + g.pad_h) % st, j0 = (pw + g.pad_w) % st;
        GemmArgs a = {};
        a.g.src = dy; a.g.Hs = g.Ho; a.g.Ws = g.Wo; a.g.Cs = g.K;
        a.g.out_h = (g.H - ph + st - 1) / st; a.g.out_w = (g.W - pw + st - 1) / st;
        a.g.mul = 1; a.g.add_y = (ph + g.pad_h - i0) / st; a.g.add_x = (pw + g.pad_w - j0) / st;
        a.g.taps_x = taps_x; a.g.taps_y = taps_y; a.g.step = -1;
        a.M = (int)(g.batch * a.g.out_h * a.g.out_w); a.N = g.C; a.K = taps_y * taps_x * g.K;
        a.g.src_bytes = (unsigned)(g.batch * g.Ho * g.Wo * g.K * 4);
        a.b.w_bytes = (unsigned)((int64_t)g.K * g.kh * g.kw * g.C * 4);
        a.b.w = w; a.b.ld = g.kh * g.kw * g.C; a.b.kc = g.K; a.b.taps_x = taps_x;
        a.b.i0 = i0; a.b.j0 = j0; a.b.si = st; a.b.kw = g.kw; a.b.c = g.C;
        a.o.out = dx; a.o.mask = mask_or_null; a.o.dense = (st == 1);
        a.o.out_bytes = (unsigned)(g.batch * g.H * g.W * g.C * 4);
        a.o.OH = g.H; a.o.OW = g.W; a.o.omul = st; a.o.oadd_y = ph; a.o.oadd_x = pw;
        a.k_per_split = round_up(a.K, BKT);
        a.trace = g_trace; a.xcd = 1;
        if (fast) {
            a.g.rmin = (a.g.add_y * g.Wo + a.g.add_x) * g.K;
            a.g.dmin = -((taps_y - 1) * g.Wo + (taps_x - 1)) * g.K;
            a.g.origin = a.g.rmin + a.g.dmin;
            a.g.src_bytes = (unsigned)(g.batch * g.Ho * g.Wo * g.K * 4 - (int64_t)a.g.origin * 4);
            a.g.mg_w = div_magic((int64_t)a.M + 256, a.g.out_w); a.g.mg_h = div_magic((int64_t)a.M + 256, a.g.out_h);
        }
        return a;
    };
    if (fast) {
        GemmArgs a = describe(0, 0);
        if (st > 1) {                       // all parity classes in one launch (blockIdx.z = class)
            int n = 0, max_m = 0;
            for (int ph = 0; ph < st && ph < g.H; ++ph)
                for (int pw = 0; pw < st && pw < g.W; ++pw) {
                    const GemmArgs q = describe(ph, pw);
                    GemmArgs::Parity& e = a.par[n++];
                    e.M = q.M; e.out_h = q.g.out_h; e.out_w = q.g.out_w; e.add_y = q.g.add_y; e.add_x = q.g.add_x;
                    e.mg_w = q.g.mg_w; e.mg_h = q.g.mg_h;
                    e.rmin = q.g.rmin; e.dmin = q.g.dmin; e.origin = q.g.origin; e.src_bytes = q.g.src_bytes;
                    e.i0 = q.b.i0; e.j0 = q.b.j0; e.oadd_y = q.o.oadd_y; e.oadd_x = q.o.oadd_x;
                    if (q.M > max_m) max_m = q.M;
                }
            a.n_par = n; a.M = max_m;       // grid covers the largest class; smaller ones exit early
            a.xcd = 0;                      // (classes of one image region are far apart in id: contiguous ranges measured slower)
        }
        // Dense layers at small batch (the DQN updates: 32 rows; or a narrow input such as the C51 head's 256):
        // a handful of 128x128 tiles walking the whole reduction is a latency chain (2 workgroups x 36 k-tiles:
        // 83 us whatever the batch); 64x64 tiles with the reduction split as in the forward pass, rectifier
        // mask applied by the fold.
        const bool small = st == 1 && a.N > 64 && (int64_t)a.M * a.N <= ((int64_t)1 << 20) && a.N % 4 == 0;
        if (plan_only) {
            plan_only->a = a; plan_only->fast = true; plan_only->has_pad = has_pad;
            plan_only->cfg = a.N <= 32 ? 0 : a.N <= 64 ? 1 : small ? 3 : 2;
            return 0;
        }
        if (small) {
            int splits = 1, per = a.k_per_split;
            if (workspace) plan_split(((a.M + 63) / 64) * ((a.N + 63) / 64), a.K, &splits, &per, 3 * TARGET_WGS);
            if ((int64_t)splits * a.M * a.N * 4 > workspace_bytes) { splits = 1; per = round_up(a.K, BKT); }
            a.k_per_split = per;
            if (splits > 1) {
                a.o.out = (float*)workspace; a.o.mask = nullptr; a.split_stride = (int64_t)a.M * a.N;
            }
            if (g_split) rc = launch_igemm_split<2, 2, 1, 1, FBK, false, false, 3>(a, false, has_pad, s, splits);
            else rc = launch_igemm<2, 2, 1, 1, FBK, false>(a, splits, false, has_pad, s);
            if (rc || splits == 1) return rc;
            return launch_fold((const float*)workspace, splits, (int64_t)a.M * a.N, nullptr, 4, 0, dx, s, mask_or_null);
        }
        if (wt && g_split && a.N > 16 && a.N <= 64 && g_dgrad_wt) {
            // k-contiguous weights: the forward pass's kernels on the data gradient's gather (B_KC = true)
            a.b.w = wt; a.b.ld = a.K; a.b.cls = a.N * a.K;
            const int tiles = ((a.M + 127) / 128) * (a.n_par ? a.n_par : 1);
            const int v = a.N <= 32 || ((g_fwd_tile >= 0 && g_fwd_tile <= 2) ? (g_fwd_tile == 1) : (2 * tiles <= TARGET_WGS));
            if (v) rc = launch_igemm_split<4, 1, 1, 1, FBK, true, false, 2>(a, false, has_pad, s);
            else rc = launch_igemm_split<4, 1, 1, 2, FBK, true, false, 2>(a, false, has_pad, s);
            return rc;
        }
        if (a.N <= 16) rc = launch_igemm<4, 1, 2, 1, 16, false, true>(a, 1, false, has_pad, s);      // 16-wide MFMA tiles
        else if (g_split && a.N <= 32) rc = launch_igemm_split<4, 1, 1, 1, FBK, false, false, 2>(a, false, has_pad, s);
        else if (g_split && a.N <= 64) {
            // (as in the forward: the column split while every half tile gets a CU of its own; same products, same order)
            const int tiles = ((a.M + 127) / 128) * (a.n_par ? a.n_par : 1);
            const int v = (g_fwd_tile >= 0 && g_fwd_tile <= 2) ? (g_fwd_tile == 1) : (2 * tiles <= TARGET_WGS);
            if (v) rc = launch_igemm_split<4, 1, 1, 1, FBK, false, false, 2>(a, false, has_pad, s);
            else rc = launch_igemm_split<4, 1, 1, 2, FBK, false, false, 2>(a, false, has_pad, s);
        }
        else if (g_split) rc = launch_igemm_split<2, 2, 2, 2, FBK, false, false, 1>(a, false, has_pad, s);
        // 17 .. 32 columns: 64x32 tiles on 16-wide MFMAs at five waves per SIMD (3 800 tiles instead of 1 900 of 128 rows
        // for the stride-2 gradient of the PPO minibatch: 52.5 -> 49.5 us; 32- and 96-row tiles, six waves: no better)
        else if (a.N <= 32 && a.N > 16) rc = launch_igemm_occ<2, 2, 2, 1, 32, false, true, 5>(a, false, has_pad, s);
        else if (a.N <= 32) rc = launch_igemm<4, 1, 1, 1, 16, false>(a, 1, false, has_pad, s);      // 16-wide k-tile: see forward
        else if (a.N <= 64) rc = launch_n64<false>(a, false, has_pad, s);
        else rc = launch_igemm<2, 2, 2, 2, FBK, false>(a, 1, false, has_pad, s);
        return rc;
    }
    if (plan_only) { plan_only->fast = false; return 0; }
    for (int ph = 0; ph < st && ph < g.H; ++ph) {
        for (int pw = 0; pw < st && pw < g.W; ++pw) {
            const GemmArgs a = describe(ph, pw);
            const bool uni = g.K % 16 == 0;                 // a 16-wide k-tile never straddles two filter taps
            if (a.N <= 32) rc = uni ? launch_rowgather<4, 1, 2, 1, 16, false, true>(a, 1, s) : launch_rowgather<4, 1, 2, 1, 16, false, false>(a, 1, s);
            else if (a.N <= 64) rc = uni ? launch_rowgather<2, 2, 2, 1, 16, false, true>(a, 1, s) : launch_rowgather<2, 2, 2, 1, 16, false, false>(a, 1, s);
            else rc = uni ? launch_rowgather<2, 2, 2, 2, 16, false, true>(a, 1, s) : launch_rowgather<2, 2, 2, 2, 16, false, false>(a, 1, s);
            if (rc) return rc;
        }
    }
    return 0;
}

// weight gradient; with more than one row split the partials go to `workspace` and *splits_out > 1
// bias_part_out (optional): the fast kernels also leave [splits][K_out] column sums of dy behind the
// weight partials in `workspace`; *bias_part_out = their address, or null when the generic kernels ran.
int wgrad_impl(const float* dy, const float* x, float* dw, const arl_conv_geom* geom, void* workspace,
               int64_t workspace_bytes, int* splits_out, int64_t* total_out, float** bias_part_out,
               WgradPlan* plan_only, void* stream) {
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
    a.trace = g_trace; a.xcd = 1;
    int bm, bn;
    if (g.K <= 16) { bm = 16; bn = 128; }
    else if (g.K <= 32) { bm = 32; bn = 128; }
    else if (g.K <= 64) { bm = 64; bn = 64; }
    else { bm = 128; bn = 128; }
    const int tiles = ((a.K_out + bm - 1) / bm) * ((a.N + bn - 1) / bn);
    int splits, per;
    // Row splits: the 128x128 configuration is MFMA-bound at one workgroup per CU (and its partials are
    // large); the small-tile ones need two to three resident workgroups per CU to cover their barrier
    // and load latencies (measured at the PPO minibatch: 256 -> 512 workgroups, 57 -> 50 us).
    plan_split(tiles, a.Mred, &splits, &per, g.K <= 32 ? 2 * TARGET_WGS : g.K <= 64 ? 3 * TARGET_WGS : TARGET_WGS);
    a.m_per_split = per;
    const int64_t total = (int64_t)a.K_out * a.N;
    if (splits > 1) {
        ARL_REQUIRE((int64_t)splits * total * 4 <= workspace_bytes, ARL_E_RANGE, "workspace too small");
        a.part = (float*)workspace;
    } else {
        a.part = dw;
    }
    constexpr int FBK = 32;
    const bool has_pad = g.pad_h > 0 || g.pad_w > 0;
    const bool fast = !g_force_generic && a.Mred % FBK == 0 && per % FBK == 0 && (!has_pad || g.kh * g.kw <= 32);
    if (bias_part_out) *bias_part_out = nullptr;
    if (fast && bias_part_out) {
        const int64_t used = splits > 1 ? (int64_t)splits * total : 0;
        ARL_REQUIRE((used + (int64_t)splits * a.K_out) * 4 <= workspace_bytes, ARL_E_RANGE, "workspace too small");
        a.bias_part = (float*)workspace + used;             // total % 4 == 0: stays 16-byte aligned
        *bias_part_out = a.bias_part;
    }
    if (fast) {
        a.g.taps_y = g.kh; a.g.dmin = 0;
        a.g.rmin = (a.g.add_y * g.W + a.g.add_x) * g.C;
        a.g.origin = a.g.rmin;
        a.g.src_bytes = (unsigned)(g.batch * g.H * g.W * g.C * 4 - (int64_t)a.g.origin * 4);
        const int img = g.Ho * g.Wo;
        a.adv_b = WG_ROWS / img;
        a.adv_y = (WG_ROWS % img) / g.Wo;
        a.adv_x = (WG_ROWS % img) % g.Wo;
        if (plan_only) {
            plan_only->a = a; plan_only->fast = true; plan_only->has_pad = has_pad;
            plan_only->cfg = g.K <= 32 ? 0 : g.K <= 64 ? 1 : 2;
            plan_only->splits = splits; plan_only->total = total;
            return 0;
        }
        if (g.K <= 16) rc = launch_wgrad_fast<1, 4, 1, 1, FBK, true>(a, splits, has_pad, s);       // 16-row MFMA tiles
        else if (g_split && g.K <= 32) rc = launch_wgrad_split<1, 4, 1, 1, FBK, false, 2>(a, splits, has_pad, s);
        else if (g_split && g.K <= 64) rc = launch_wgrad_split<2, 2, 1, 1, FBK, false, 2>(a, splits, has_pad, s);
        else if (g_split) rc = launch_wgrad_split<2, 2, 2, 2, FBK, false, 1>(a, splits, has_pad, s);
        else if (g.K <= 32) rc = launch_wgrad_fast<1, 4, 1, 1, FBK>(a, splits, has_pad, s);
        else if (g.K <= 64) rc = launch_wgrad_fast<2, 2, 1, 1, FBK>(a, splits, has_pad, s);
        else rc = launch_wgrad_fast<2, 2, 2, 2, FBK>(a, splits, has_pad, s);
    } else {
        if (plan_only) { plan_only->fast = false; return 0; }
        if (g.K <= 32) rc = launch_wgrad<1, 4, 1, 1, 16>(a, splits, s);
        else if (g.K <= 64) rc = launch_wgrad<2, 2, 1, 1, 16>(a, splits, s);
        else rc = launch_wgrad<2, 2, 2, 2, 16>(a, splits, s);
    }
    *splits_out = splits;
    *total_out = total;
    return rc;
}
}  // namespace

extern "C" int arl_conv2d_bwd_data(const float* dy, const float* w, const float* wt_or_null, const float* mask_or_null,
                                   float* dx, const arl_conv_geom* geom, const arl_corun_job* job_or_null,
                                   int32_t* job_taken_or_null, void* stream) {
    ARL_ROUTE_SCOPE(geom, job_or_null);
    const int rc = dgrad_impl(dy, w, mask_or_null, dx, geom, nullptr, stream, nullptr, 0, wt_or_null);
    if (job_taken_or_null) *job_taken_or_null = t_ctx.corun_taken ? 1 : 0;
    return rc;
}

extern "C" int arl_conv2d_bwd_weight(const float* dy, const float* x, float* dw, const arl_conv_geom* geom,
                                     void* workspace, void* stream) {
    ARL_ROUTE_SCOPE(geom, nullptr);
    int splits = 1;
    int64_t total = 0;
    int rc = wgrad_impl(dy, x, dw, geom, workspace, arl_conv_workspace_bytes(), &splits, &total, nullptr, nullptr, stream);
    if (rc || splits == 1) return rc;
    return launch_fold((const float*)workspace, splits, total, nullptr, 4, 0, dw, (hipStream_t)stream);
}

namespace {
void bias_item(arl_fold_item* item, const float* bias_part, float* dbias, int splits, int channels) {
    item->part = bias_part; item->out = dbias; item->total = channels;
    item->splits = bias_part ? splits : -1;             // -1: not produced (generic kernels ran)
    item->valid = 0;
}
int wgrad_parts_impl(const float* dy, const float* x, float* dw, const arl_conv_geom* geom,
                     void* workspace, int64_t workspace_bytes, arl_fold_item* item,
                     float* dbias_or_null, arl_fold_item* bias_item_or_null, void* stream) {
    ARL_REQUIRE(item && (!dbias_or_null || bias_item_or_null), ARL_E_ARG, "null pointer");
    ARL_REQUIRE(!dbias_or_null || arl::aligned16(dbias_or_null), ARL_E_ALIGN, "16-byte alignment");
    int splits = 1;
    int64_t total = 0;
    float* bias_part = nullptr;
    int rc = wgrad_impl(dy, x, dw, geom, workspace, workspace_bytes, &splits, &total,
                        dbias_or_null ? &bias_part : nullptr, nullptr, stream);
    item->part = (const float*)workspace; item->out = dw; item->total = total;
    item->splits = splits > 1 ? splits : 0;             // 0: dw is already final
    item->valid = 0;
    if (dbias_or_null) bias_item(bias_item_or_null, bias_part, dbias_or_null, splits, geom->out_c);
    return rc;
}
}  // namespace

extern "C" int arl_conv2d_bwd_weight_parts(const float* dy, const float* x, float* dw, const arl_conv_geom* geom,
                                           void* workspace, int64_t workspace_bytes, arl_fold_item* item,
                                           float* dbias_or_null, arl_fold_item* bias_item_or_null, void* stream) {
    ARL_ROUTE_SCOPE(geom, nullptr);
    return wgrad_parts_impl(dy, x, dw, geom, workspace, workspace_bytes, item, dbias_or_null, bias_item_or_null, stream);
}

// ------------------------------------------------------------------------------------------
// Convolution 1 straight from the sampler's observations: planar u8 [rows][C][H][W] read in place
// (optionally through a row-index list), weights / weight gradient in (K, C, kh, kw) order.
// ------------------------------------------------------------------------------------------
namespace {
struct U8Geom { int H, W, C, K, kh, kw, stride, Ho, Wo; int64_t batch; };

int check_u8(const uint8_t* obs, int64_t obs_rows, const arl_conv_geom* g, U8Geom* o) {
    ARL_REQUIRE(obs && g && obs_rows > 0, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(g->batch > 0 && g->in_h > 0 && g->in_w > 0 && g->in_c > 0 && g->out_c > 0 && g->kh > 0 && g->kw > 0 &&
                    g->stride > 0, ARL_E_ARG, "conv: bad geometry");
    const int cpr = g->kw / 4;
    ARL_REQUIRE(g->pad_h == 0 && g->pad_w == 0 && g->stride % 4 == 0 && g->in_w % 4 == 0 &&
                    (g->in_h * g->in_w) % 4 == 0 && (g->kw == 4 || g->kw == 8 || g->kw == 16) && g->kh % (4 / cpr) == 0 &&
                    g->out_c % 4 == 0 && g->out_c <= 32 && ((uintptr_t)obs & 3) == 0,
                ARL_E_RANGE, "u8 conv: needs pad 0, stride % 4 == 0, width % 4 == 0, kw in {4, 8, 16}, <= 32 filters");
    o->H = g->in_h; o->W = g->in_w; o->C = g->in_c; o->K = g->out_c; o->kh = g->kh; o->kw = g->kw; o->stride = g->stride;
    o->Ho = (g->in_h - g->kh) / g->stride + 1; o->Wo = (g->in_w - g->kw) / g->stride + 1; o->batch = g->batch;
    const int64_t lim = (int64_t)OOB / 4;
    ARL_REQUIRE(o->Ho > 0 && o->Wo > 0 && g->batch * (int64_t)o->Ho * o->Wo * g->out_c < lim &&
                    obs_rows * (int64_t)g->in_c * g->in_h * g->in_w < (int64_t)OOB,
                ARL_E_RANGE, "conv: tensor larger than the 32-bit buffer offsets address");
    return 0;
}

void fill_u8(GatherDesc* d, const uint8_t* obs, int64_t obs_rows, const int32_t* idx, float scale, const U8Geom& g) {
    d->src8 = obs; d->idx = idx; d->scale = scale;
    d->plane = g.H * g.W; d->img_bytes = g.C * g.H * g.W; d->kh8 = g.kh; d->kw8 = g.kw;
    d->src_bytes = (unsigned)(obs_rows * d->img_bytes);
    d->Hs = g.H; d->Ws = g.W; d->Cs = 1; d->out_h = g.Ho; d->out_w = g.Wo;
    d->mul = g.stride; d->add_y = 0; d->add_x = 0; d->taps_x = g.kw; d->taps_y = g.kh; d->step = 1;
}
}  // namespace

extern "C" int arl_conv2d_u8_fwd(const uint8_t* obs, int64_t obs_rows, const int32_t* idx_or_null, float scale,
                                 const float* w, const float* bias_or_null, float* y, const arl_conv_geom* geom,
                                 int32_t relu, void* stream) {
    ARL_REQUIRE(w && y, ARL_E_ARG, "null pointer");
    U8Geom g;
    int rc = check_u8(obs, obs_rows, geom, &g);
    if (rc) return rc;
    ARL_ROUTE_SCOPE(geom, nullptr);
    ARL_REQUIRE(arl::aligned16(w) && arl::aligned16(y) && (!bias_or_null || arl::aligned16(bias_or_null)), ARL_E_ALIGN,
                "16-byte alignment");
    if (!g_trace) {                                 // 32 filters of 8 x 8: the whole image in LDS (img_conv.hip)
        rc = launch_conv1_img(obs, obs_rows, idx_or_null, scale, w, bias_or_null, y, g.batch, g.C, g.H, g.W, g.K, g.kh, g.kw, g.stride,
                              g.Ho, g.Wo, relu, (hipStream_t)stream);
        if (rc >= 0) return rc;
    }
    GemmArgs a = {};
    fill_u8(&a.g, obs, obs_rows, idx_or_null, scale, g);
    a.M = (int)(g.batch * g.Ho * g.Wo); a.N = g.K; a.K = g.C * g.kh * g.kw;
    a.b.w = w; a.b.ld = a.K; a.b.w_bytes = (unsigned)((int64_t)a.N * a.K * 4);
    a.o.dense = 1; a.o.out = y; a.o.bias = bias_or_null; a.o.relu = relu;
    a.o.out_bytes = (unsigned)((int64_t)a.M * a.N * 4);
    a.k_per_split = a.K;                            // K % 16 == 0 by check_u8 (whole filter rows per k-tile)
    a.g.mg_w = div_magic((int64_t)a.M + 256, a.g.out_w); a.g.mg_h = div_magic((int64_t)a.M + 256, a.g.out_h);
    a.trace = g_trace; a.xcd = 1;
    constexpr int BK = 16, BM = 128;
    const dim3 grid((a.M + BM - 1) / BM, 1, 1);
    if (a.N <= 16) {
        const size_t lds = (size_t)2 * (BM * (BK + 4) + 16 * (BK + 4)) * sizeof(float);
        hipLaunchKernelGGL((igemm_u8_kernel<4, 1, 2, 1, BK, true>), grid, dim3(256), lds, (hipStream_t)stream, a);
    } else if (g_split && g.kh % (8 / (g.kw >> 2)) == 0) {
        // bf16-split products: the pixels are exact in one bf16 plane (three products), 32-deep k-tiles = whole filter rows
        return launch_igemm_split<4, 1, 2, 1, 32, true, true, 2>(a, false, false, (hipStream_t)stream);
    } else if (g.kh % (8 / (g.kw >> 2)) == 0) {
        // 17 .. 32 filters: 64x32 tiles on 16-wide MFMAs, 32-deep k-tiles (whole filter rows: kh % (32 / kw) == 0), five
        // waves per SIMD -- 3 800 tiles for the PPO minibatch instead of 1 900 of 128 rows: 52.1 -> 49.7 us
        constexpr int PBM = 64, PBK = 32;
        const dim3 pgrid((a.M + PBM - 1) / PBM, 1, 1);
        const arl::OptSeg none = {};
        const size_t lds = (size_t)2 * (PBM * (PBK + 4) + 32 * (PBK + 4)) * sizeof(float);
        hipLaunchKernelGGL((igemm_occ_kernel<2, 2, 2, 1, PBK, true, false, false, true, 5, false, true>), pgrid, dim3(256),
                           lds, (hipStream_t)stream, a, none);
        return arl::check_launch("igemm_occ_kernel (u8)");
    } else {
        const size_t lds = (size_t)2 * (BM * (BK + 4) + 32 * (BK + 4)) * sizeof(float);
        hipLaunchKernelGGL((igemm_u8_kernel<4, 1, 1, 1, BK, false>), grid, dim3(256), lds, (hipStream_t)stream, a);
    }
    return arl::check_launch("igemm_u8_kernel");
}

extern "C" int arl_conv2d_u8_bwd_weight_parts(const float* dy, const uint8_t* obs, int64_t obs_rows,
                                              const int32_t* idx_or_null, float scale, float* dw,
                                              const arl_conv_geom* geom, void* workspace, int64_t workspace_bytes,
                                              arl_fold_item* item, float* dbias_or_null,
                                              arl_fold_item* bias_item_or_null, void* stream) {
    ARL_REQUIRE(dy && dw && workspace && item && (!dbias_or_null || bias_item_or_null), ARL_E_ARG, "null pointer");
    U8Geom g;
    int rc = check_u8(obs, obs_rows, geom, &g);
    if (rc) return rc;
    ARL_ROUTE_SCOPE(geom, nullptr);
    ARL_REQUIRE(arl::aligned16(dy) && arl::aligned16(dw) && arl::aligned16(workspace) &&
                    (!dbias_or_null || arl::aligned16(dbias_or_null)), ARL_E_ALIGN, "16-byte alignment");
    WgradArgs a = {};
    a.dy = dy;
    fill_u8(&a.g, obs, obs_rows, idx_or_null, scale, g);
    a.K_out = g.K; a.N = g.C * g.kh * g.kw; a.Mred = (int)(g.batch * g.Ho * g.Wo);
    a.dy_bytes = (unsigned)((int64_t)a.Mred * g.K * 4);
    a.trace = g_trace; a.xcd = 1;
    // split route, 256 columns (spec 1: 4 planes x 8 x 8): ONE column tile of 256 -- a wave owns two 32-column tiles, so
    // every dy tile is loaded and split once per 256 columns instead of once per 128 -- at twice the k-splits (the same
    // 512 workgroups, the same partials to fold): 34.7 -> 33.1 us with its fold (profiles/r05/conv1_wgrad_wide_ab.txt;
    // 256 workgroups 37.5, 768 / 1 024 slower again)
    const bool wide = g.K > 16 && g_split && g.kw % 8 == 0 && a.N % 256 == 0;
    const int bm = g.K <= 16 ? 16 : 32, bn = wide ? 256 : 128;
    const int tiles = ((a.K_out + bm - 1) / bm) * ((a.N + bn - 1) / bn);
    int splits, per;
    // residency: the 16-row tiles (<= 16 filters) are small enough for 4 workgroups per CU (A2C-1024 batch:
    // 300 -> 242 us; 6 or 8 per CU: 248-253); the 32-row ones stay at 2 (3: same, 4: slower)
    plan_split(tiles, a.Mred, &splits, &per, (g.K <= 16 ? 4 : 2) * TARGET_WGS);
    a.m_per_split = per;
    const int64_t total = (int64_t)a.K_out * a.N;
    const int64_t used = splits > 1 ? (int64_t)splits * total : 0;
    ARL_REQUIRE((used + (int64_t)splits * a.K_out) * 4 <= workspace_bytes, ARL_E_RANGE, "workspace too small");
    a.part = splits > 1 ? (float*)workspace : dw;
    if (dbias_or_null) a.bias_part = (float*)workspace + used;
    const int img = g.Ho * g.Wo;
    a.adv_b = WG_ROWS / img; a.adv_y = (WG_ROWS % img) / g.Wo; a.adv_x = (WG_ROWS % img) % g.Wo;
    constexpr int BK = 32;
    const size_t lds = (size_t)2 * BK * (bm + bn) * sizeof(float);
    const dim3 grid((a.N + bn - 1) / bn, (a.K_out + bm - 1) / bm, splits);
    if (g.K > 16 && g_split && g.kw % 8 == 0) {     // (the split kernel gathers whole 8-pixel filter rows)
        if (wide) rc = launch_wgrad_split<1, 4, 1, 2, BK, true, 2>(a, splits, false, (hipStream_t)stream);
        else rc = launch_wgrad_split<1, 4, 1, 1, BK, true, 2>(a, splits, false, (hipStream_t)stream);
        if (rc) return rc;
    } else if (g.K <= 16) hipLaunchKernelGGL((wgrad_u8_kernel<1, 4, 1, 1, BK, true>), grid, dim3(256), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((wgrad_u8_kernel<1, 4, 1, 1, BK, false>), grid, dim3(256), lds, (hipStream_t)stream, a);
    item->part = (const float*)workspace; item->out = dw; item->total = total;
    item->splits = splits > 1 ? splits : 0;
    item->valid = 0;
    if (dbias_or_null) bias_item(bias_item_or_null, a.bias_part, dbias_or_null, splits, g.K);
    return arl::check_launch("wgrad_u8_kernel");
}


extern "C" int arl_conv2d_bwd_pair(const float* dy, const float* w, const float* wt_or_null, const float* mask_or_null,
                                   float* dx, const float* x, float* dw, const arl_conv_geom* geom, void* workspace,
                                   int64_t workspace_bytes, arl_fold_item* item, float* dbias_or_null,
                                   arl_fold_item* bias_item_or_null, const arl_corun_job* job_or_null,
                                   int32_t* job_taken_or_null, void* stream) {
    ARL_REQUIRE(item && geom && (!dbias_or_null || bias_item_or_null), ARL_E_ARG, "null pointer");
    if (job_taken_or_null) *job_taken_or_null = 0;
    ARL_ROUTE_SCOPE(geom, job_or_null);
    ARL_REQUIRE(!dbias_or_null || arl::aligned16(dbias_or_null), ARL_E_ALIGN, "16-byte alignment");
    DgradPlan dp = {};
    WgradPlan wp = {};
    int splits = 1;
    int64_t total = 0;
    float* bias_part = nullptr;
    int rc = dgrad_impl(dy, w, mask_or_null, dx, geom, &dp, stream);
    if (rc) return rc;
    rc = wgrad_impl(dy, x, dw, geom, workspace, workspace_bytes, &splits, &total,
                    dbias_or_null ? &bias_part : nullptr, &wp, stream);
    if (rc) return rc;
    const bool has_pad = dp.has_pad || wp.has_pad;
    // Only the widest tile configuration pairs up: a shared launch runs every workgroup at the larger of the
    // two LDS / register footprints, which costs the smaller-tile weight-gradient kernels their occupancy
    // (measured: conv 2 / conv 3 pairs 9-18 us slower than apart, the dense pair 11 us faster).
    // ... and only while the data gradient has at most two 128 x 128 tiles per CU: beyond that the shared launch falls
    // behind the two separate ones, and badly (spec-0 dense at the A2C-1024 batch, 5 120 x 3 456 x 256: 380.6 us paired,
    // 213.5 apart, weight-gradient fold included; 6 912 x 512 at 4 096 rows: 1 393 against 569; the PPO minibatch's 216
    // tiles: 63.3 against 79.3 -- profiles/r06/pair_size_probe.txt)
    const int64_t dgrad_tiles = (int64_t)((dp.a.M + 127) / 128) * ((dp.a.N + 127) / 128) * (dp.a.n_par ? dp.a.n_par : 1);
    const bool paired = dp.fast && wp.fast && dp.cfg == 2 && wp.cfg == 2 && dgrad_tiles <= 2 * TARGET_WGS &&
                        (!has_pad || geom->kh * geom->kw <= 32) && !g_trace;
    if (!paired) {
        // the data gradient may split its reduction: it gets the upper half of the workspace (and is folded at
        // once), the weight gradient's deferred partials the lower half
        const int64_t half = (workspace_bytes / 2) & ~(int64_t)15;
        rc = dgrad_impl(dy, w, mask_or_null, dx, geom, nullptr, stream, (char*)workspace + half, workspace_bytes - half,
                        wt_or_null);
        if (job_taken_or_null) *job_taken_or_null = t_ctx.corun_taken ? 1 : 0;
        if (rc) return rc;
        return wgrad_parts_impl(dy, x, dw, geom, workspace, half, item, dbias_or_null, bias_item_or_null, stream);
    }
    hipStream_t s = (hipStream_t)stream;
    // split kernels: 16-deep k-tiles halve the LDS images (98 -> 49 KB), so that TWO workgroups share a CU and one's
    // loads / splits / LDS stores run under the other's MFMAs (one 128x128 workgroup alone keeps the matrix pipe ~50 %
    // busy): the dense pair of the PPO minibatch 89.4 -> 71.9 us, bit-identical
    if (g_split) rc = launch_pair<2, 2, 2, 2, 2, 2, 2, 2, 16>(dp, wp, has_pad, s);
    else rc = launch_pair<2, 2, 2, 2, 2, 2, 2, 2>(dp, wp, has_pad, s);
    item->part = (const float*)workspace; item->out = dw; item->total = wp.total;
    item->splits = wp.splits > 1 ? wp.splits : 0;
    item->valid = 0;
    if (dbias_or_null) bias_item(bias_item_or_null, bias_part, dbias_or_null, wp.splits, geom->out_c);
    return rc;
}

namespace {
__global__ __launch_bounds__(256) void dgrad_weights_kernel(const arlw::DgradWtArgs a) {
    arlw::dgrad_wt_block(a, (int)blockIdx.x, (int)threadIdx.x);
}
}  // namespace

namespace arlw {
int dgrad_wt_plan(const arl_dgrad_wt* items, int32_t n, DgradWtArgs* out, int* blocks_out) {
    ARL_REQUIRE(items && n > 0 && n <= ARL_DGRAD_WT_MAX, ARL_E_ARG, "1 .. ARL_DGRAD_WT_MAX items");
    DgradWtArgs& a = *out;
    a = DgradWtArgs{};
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        arlc::Geom g;
        int rc = arlc::check_geom(items[i].geom, &g);
        if (rc) return rc;
        ARL_REQUIRE(items[i].w && items[i].wt && items[i].w != items[i].wt, ARL_E_ARG, "null / aliased pointer");
        ARL_REQUIRE(arl::aligned16(items[i].w) && arl::aligned16(items[i].wt), ARL_E_ALIGN, "16-byte alignment");
        const int st = g.stride;
        ARL_REQUIRE(g.kh % st == 0 && g.kw % st == 0 && st * st <= 4 && g.H >= st && g.W >= st, ARL_E_RANGE,
                    "kernel size divisible by the stride, stride <= 2");
        DgradWtItem& q = a.it[i];
        q.w = items[i].w; q.wt = items[i].wt; q.K = g.K; q.kh = g.kh; q.kw = g.kw; q.C = g.C; q.st = st;
        q.taps_x = g.kw / st; q.kred = (g.kh / st) * (g.kw / st) * g.K; q.total = g.K * g.kh * g.kw * g.C;
        for (int ph = 0; ph < st; ++ph)
            for (int pw = 0; pw < st; ++pw) {
                q.i0[ph * st + pw] = (ph + g.pad_h) % st;
                q.j0[ph * st + pw] = (pw + g.pad_w) % st;
            }
        a.block_start[i] = blocks;
        blocks += (q.total + 255) / 256;
    }
    a.block_start[n] = blocks; a.n = n;
    *blocks_out = blocks;
    return 0;
}
}  // namespace arlw

extern "C" int arl_conv2d_dgrad_weights(const arl_dgrad_wt* items, int32_t n, void* stream) {
    arlw::DgradWtArgs a;
    int blocks = 0;
    int rc = arlw::dgrad_wt_plan(items, n, &a, &blocks);
    if (rc) return rc;
    hipLaunchKernelGGL(dgrad_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return arl::check_launch("dgrad_weights_kernel");
}

extern "C" int arl_fold_many(const arl_fold_item* items, int32_t n, void* stream) {
    ARL_REQUIRE(items || n == 0, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(n >= 0 && n <= ARL_FOLD_MAX_ITEMS, ARL_E_RANGE, "too many fold items");
    FoldManyArgs a = {};
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        if (items[i].splits <= 0) continue;
        ARL_REQUIRE(items[i].part && items[i].out && items[i].total > 0 && (items[i].total & 3) == 0, ARL_E_ARG,
                    "fold item: null pointer or length not a multiple of 4");
        ARL_REQUIRE(arl::aligned16(items[i].part) && arl::aligned16(items[i].out), ARL_E_ALIGN, "16-byte alignment");
        a.items[a.n] = items[i];
        a.block_start[a.n++] = blocks;
        blocks += fold_blocks(items[i].splits, items[i].total >> 2);
    }
    if (a.n == 0) return 0;
    a.block_start[a.n] = blocks;
    a.wide_from = g_fold_wide;
    hipLaunchKernelGGL(fold_many_kernel, dim3((unsigned)blocks), dim3(FOLD_THREADS), 0, (hipStream_t)stream, a);
    return arl::check_launch("fold_many_kernel");
}
