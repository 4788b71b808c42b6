// Prototype (round 3): dense forward y[M][N] = relu(x[M][K] . W[N][K]^T + b) on the bf16 matrix pipe with exact three-way
// bf16 splits, the WEIGHTS pre-split once (they change once per optimiser step, not once per launch) and stored in
// MFMA-fragment order: a wave's weight fragment is ONE fully coalesced 1 KB load straight into registers -- no LDS, no
// split, no barrier for that operand.  The activations pass through LDS: per round a 512-thread workgroup loads a
// 128 x 32 fp32 block (whole 128-byte lines), a thread splits the eight values of ONE fragment lane and writes them where
// the MFMA reads them (3 ds_write_b128; a wave's fragment read is 1 KB contiguous: conflict-free).
//   grid: 16 output tiles of 128 x 128 x 16 K-splits = 256 workgroups (one per CU), 8 waves = two groups of 2 x 2 waves
//   (64 x 64 each); the groups take alternate 16-k steps of the split and add their accumulators through LDS at the end.
// Checks against a float64 contraction and times it.  usage: dense_fwd_proto [M [rotate]]   (-DKO=..: knock-outs)
// MEASURED (MI355X, 512 x 512 x 6912, 20 launches in a graph): 38-41 us (+ 3.7 us fold, + 8.2 us weight prep per optimiser
// step) against 39.5 + 5.5 us in production: correct (2.3e-7 of the float64 reference), NOT faster, NOT adopted.  Per wave
// the k-loop takes 55-60 k cycles for 32.3 k cycles of matrix-pipe time, and NO schedule moved that: weight / activation
// loads 1, 2 or 3 rounds ahead (54.4-56.2 k), fragment reads a round ahead with three LDS stages (59.9 k), the two waves
// of a SIMD in opposite phases with one barrier per round (59.9 k) or strictly ping-ponged between a compute and a load
// segment with two (59.8 k: the version below), the four workgroups that share a slab starting from different rounds
// (56.8 k).  Knock-outs of the one-barrier version (results wrong, timing only): MFMAs alone 32.9 k (= the pipe's own
// time: 100 %), + weight loads 41.2 k, + activation staging 44.6 k, both 55-56 k; without the barrier 47.0 k; everything
// BUT the MFMAs 16.5 k; of the ping-pong version: MFMAs + fragment reads + barriers 37.4 k, + activation staging
// 45.6 k, + weight loads (no staging) 51.1 k -- six 1 KB loads in the OTHER wave's segment still cost the slot ~490 cycles.
// The parts add up instead of overlapping -- a wave's six 1 KB weight loads cost ~90 cycles of
// its SIMD each, its staging pass ~390 -- on a SIMD whose matrix pipe runs at 836 TF/s over the launch: the level the
// production kernels are at (0.8-1.0 PF/s), and 70 % of what the CDNA4 guide reports for its best plain-HIP bf16 GEMM
// at 8192^3 (1.16-1.22 PF/s = 0.48 of the 2.5 PF/s peak).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

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

// fragment-ordered, pre-split weights: wf[s][j][p][lane] (16 bytes each): lane (l31, half) of 16-k step s, column tile j
// holds W[32 j + l31][16 s + 8 half .. + 7], piece p (0 = h, 1 = m, 2 = l)
__global__ void wprep_kernel(const float* w, u32x4* wf, int N, int K) {
    const int NJ = N / 32, total = (K / 16) * NJ * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63, j = (i >> 6) % NJ, s = (i >> 6) / NJ;
        const float* src = w + (size_t)(32 * j + (lane & 31)) * K + 16 * s + 8 * (lane >> 5);
        unsigned h[4], m[4], l[4];
        for (int q = 0; q < 4; ++q) split_pair(src[2 * q], src[2 * q + 1], h[q], m[q], l[q]);
        u32x4* dst = wf + ((size_t)(s * NJ + j) * 3) * 64 + lane;
        dst[0] = u32x4{h[0], h[1], h[2], h[3]}; dst[64] = u32x4{m[0], m[1], m[2], m[3]}; dst[128] = u32x4{l[0], l[1], l[2], l[3]};
    }
}

struct DArgs {
    const float* x; const u32x4* wf; float* part;
    int M, N, K, splits, steps, rotate; // steps: 16-k steps per split
    unsigned long long* trace;
};

constexpr int NT = 512;
#ifndef KO
#define KO 0           // knock-outs (timing only, results wrong): 1 no MFMAs, 2 no weight loads after the first, 4 no activation staging after the first, 8 no barrier
#endif
#ifndef BDEPTH
#define BDEPTH 2
#endif
#ifndef ADEPTH
#define ADEPTH 2
#endif
constexpr int STAGE = 2 * 4 * 3 * 1024;         // [group][row tile][piece] x 1 KB

template <int STEPS, int BD, int AD>      // 16-k steps per split; rounds the weight / activation loads run ahead
__global__ __launch_bounds__(NT) void dense_fwd_kernel(const DArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    // XCD-aware: the 32 workgroups of an XCD = 2 splits x 16 tiles (they share the splits' slabs of x and W in its L2)
    const int id = blockIdx.x, xcd = id & 7, in_xcd = id >> 3;
    const int per_xcd = gridDim.x >> 3, tiles = (a.M / 128) * (a.N / 128);
    const int lin = xcd * per_xcd + in_xcd;
    const int split = lin / tiles, tile = lin - split * tiles;
    const int tm = tile / (a.N / 128), tn = tile - tm * (a.N / 128);
    constexpr int nsteps = STEPS, rounds = (STEPS + 1) >> 1;
    const int NJ = a.N / 32;
    const int s0 = split * nsteps;
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if (a.trace) t0 = __builtin_readcyclecounter();
    // ---- activations: thread (row = tid / 4, c = tid % 4) owns k = 8 c .. 8 c + 7 of each round's 32-k block
    const int row = tid >> 2, c = tid & 3;
    const float* xsrc = a.x + (size_t)(tm * 128 + row) * a.K + (size_t)s0 * 16 + 8 * c;
    const unsigned wr_off = (unsigned)((((c >> 1) * 4 + (row >> 5)) * 3) * 1024 + ((row & 31) + 32 * (c & 1)) * 16);
    // activation loads run AD rounds ahead of their split + LDS store (register sets ra[block % AD]), the store one
    // round ahead of the MFMAs (two LDS stages); weight fragments run BD rounds ahead (register sets fb[round % (BD + 1)])
    float4 ra[AD][2];
    const int rot = a.rotate ? (((tm + tn) & 3) * rounds) >> 2 : 0;      // the four workgroups that share a slab of x (of W) walk it from different rounds
#define ACT(v) (((v) + rot) >= rounds ? (v) + rot - rounds : (v) + rot)
    const int a_lim = nsteps - (c >> 1);            // block b of this thread's chunk exists while 2 b < a_lim
#define A_ISSUE(b) do { const int rb_ = ACT(b); if ((b) < rounds && 2 * rb_ < a_lim) { const float4* p_ = reinterpret_cast<const float4*>(xsrc + 32 * rb_); \
                            ra[(b) % AD][0] = p_[0]; ra[(b) % AD][1] = p_[1]; } \
                        else { ra[(b) % AD][0] = make_float4(0.f, 0.f, 0.f, 0.f); ra[(b) % AD][1] = ra[(b) % AD][0]; } } while (0)
#define A_STORE(b) do { unsigned h0, h1, h2, h3, m0, m1, m2, m3, l0, l1, l2, l3; const float4 q0_ = ra[(b) % AD][0], q1_ = ra[(b) % AD][1]; \
        split_pair(q0_.x, q0_.y, h0, m0, l0); split_pair(q0_.z, q0_.w, h1, m1, l1); \
        split_pair(q1_.x, q1_.y, h2, m2, l2); split_pair(q1_.z, q1_.w, h3, m3, l3); \
        char* d_ = lds + ((b) & 1) * STAGE + wr_off; \
        *reinterpret_cast<u32x4*>(d_) = u32x4{h0, h1, h2, h3}; \
        *reinterpret_cast<u32x4*>(d_ + 1024) = u32x4{m0, m1, m2, m3}; \
        *reinterpret_cast<u32x4*>(d_ + 2048) = u32x4{l0, l1, l2, l3}; } while (0)
    const u32x4* wsrc = a.wf + ((size_t)(s0 + grp) * NJ + tn * 4 + wn * 2) * 3 * 64 + lane;
    const size_t wstep = (size_t)2 * NJ * 3 * 64;
    u32x4 fb[BD][2][3];
#define B_ISSUE(r) do { const int rr_ = ACT(r); if ((r) < rounds && 2 * rr_ + grp < nsteps) { const u32x4* p_ = wsrc + (size_t)rr_ * wstep; _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) \
        _Pragma("unroll") for (int pc_ = 0; pc_ < 3; ++pc_) fb[(r) % BD][j_][pc_] = p_[(j_ * 3 + pc_) * 64]; } } while (0)
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    // Ping-pong: the two waves of a SIMD (group 0 / group 1) alternate between a COMPUTE segment (the 36 MFMAs of one of
    // the group's 16-k steps) and a LOAD segment (weight fragments for a later step, fragment reads for the next one, split
    // + LDS store of the next block, global loads of a later block), one barrier per slot: while one group's MFMAs hold the
    // matrix pipe the other's vector-memory / vector-ALU / LDS instructions issue.  Group g computes round r in slot
    // 2 r + g; its load segment for round r is the slot before.
    u32x4 fa[2][3];
    const unsigned rd_off = (unsigned)(((grp * 4 + wm * 2) * 3) * 1024 + lane * 16);
#define LOADSEG(r) do { if (!(KO & 2)) B_ISSUE((r) + BD - 1); \
        if ((r) < rounds) { const char* st_ = lds + ((r) & 1) * STAGE + rd_off; _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) \
            _Pragma("unroll") for (int pc_ = 0; pc_ < 3; ++pc_) fa[i_][pc_] = *reinterpret_cast<const u32x4*>(st_ + (i_ * 3 + pc_) * 1024); } \
        if (!(KO & 4) && (r) + 1 < rounds) { A_STORE((r) + 1); A_ISSUE((r) + 1 + AD); } } while (0)
#define COMPUTE(r) do { if (!(KO & 1) && 2 * ACT(r) + grp < nsteps) { \
        _Pragma("unroll") for (int s_ = 4; s_ >= 0; --s_) _Pragma("unroll") for (int pa_ = 0; pa_ < 3; ++pa_) { \
            const int pb_ = s_ - pa_; if (pb_ < 0 || pb_ > 2) continue; \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) \
                acc[i_][j_] = mfma_bf16(fb[(r) % BD][j_][pb_], fa[i_][pa_], acc[i_][j_]); } } } while (0)
#pragma unroll
    for (int b = 0; b < AD; ++b) A_ISSUE(b);
#pragma unroll
    for (int r = 0; r + 1 < BD; ++r) B_ISSUE(r);
    A_STORE(0);
    A_ISSUE(AD);
    __syncthreads();
    if (grp == 0) LOADSEG(0);
    __syncthreads();
    if (a.trace) t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int s = 0; s < 2 * rounds; ++s) {
        if (grp == 0) {
            if (s % 2 == 0) COMPUTE(s / 2); else LOADSEG((s + 1) / 2);
        } else {
            if (s % 2 == 0) LOADSEG(s / 2); else COMPUTE((s - 1) / 2);
        }
        if (!(KO & 8)) __syncthreads();
    }
    if (a.trace) t2 = __builtin_readcyclecounter();
    // ---- the two groups' sums: group 1 -> LDS, group 0 adds and stores the split's partial tile
    float* red = reinterpret_cast<float*>(lds) + (wave & 3) * (4 * 16 * 64) + lane;
    if (grp == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) red[((i * 2 + j) * 16 + v) * 64] = acc[i][j][v];
    }
    __syncthreads();
    if (grp == 0) {
        const int l31 = lane & 31, half = lane >> 5;
        float* out = a.part + (size_t)split * a.M * a.N;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = tm * 128 + (wm * 2 + i) * 32 + l31;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n0 = tn * 128 + (wn * 2 + j) * 32 + 4 * half;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 v;
                    v.x = acc[i][j][4 * g + 0] + red[((i * 2 + j) * 16 + 4 * g + 0) * 64];
                    v.y = acc[i][j][4 * g + 1] + red[((i * 2 + j) * 16 + 4 * g + 1) * 64];
                    v.z = acc[i][j][4 * g + 2] + red[((i * 2 + j) * 16 + 4 * g + 2) * 64];
                    v.w = acc[i][j][4 * g + 3] + red[((i * 2 + j) * 16 + 4 * g + 3) * 64];
                    *reinterpret_cast<float4*>(out + (size_t)m * a.N + n0 + 8 * g) = v;
                }
            }
        }
    }
    if (a.trace && lane == 0) {
        unsigned long long* t = a.trace + ((size_t)blockIdx.x * 8 + wave) * 4;
        t[0] = t0; t[1] = t1; t[2] = t2; t[3] = __builtin_readcyclecounter();
    }
}

__global__ void fold_kernel(const float4* part, int splits, size_t total4, const float* bias, int N, int relu, float4* y) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    float4 s = part[i];
    for (int k = 1; k < splits; ++k) { const float4 v = part[(size_t)k * total4 + i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    const float4 b = *reinterpret_cast<const float4*>(bias + (i * 4) % N);
    s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    if (relu) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
    y[i] = s;
}

__global__ void ref_kernel(const float* x, const float* w, const float* bias, double* y, int M, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    double s = bias[n];
    for (int k = 0; k < K; ++k) s += (double)x[(size_t)m * K + k] * (double)w[(size_t)n * K + k];
    y[(size_t)m * N + n] = s > 0 ? s : 0;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 512, N = 512, K = 6912, splits = 16, steps = K / 16 / splits;
    std::vector<float> hx((size_t)M * K), hw((size_t)N * K), hb(N);
    srand(3);
    for (auto& v : hx) v = (rand() & 3) ? ((rand() / (float)RAND_MAX)) * 2.f : 0.f;          // rectified activations
    for (auto& v : hw) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.05f;
    for (auto& v : hb) v = ((rand() / (float)RAND_MAX) - 0.5f);
    float *dx, *dw, *db, *dpart, *dy; double* dref; u32x4* dwf;
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&db, N * 4));
    CK(hipMalloc(&dpart, (size_t)splits * M * N * 4)); CK(hipMalloc(&dy, (size_t)M * N * 4)); CK(hipMalloc(&dref, (size_t)M * N * 8));
    CK(hipMalloc(&dwf, (size_t)N * K * 6));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ref_kernel, dim3((N + 63) / 64, M), dim3(64), 0, 0, dx, dw, db, dref, M, N, K);
    hipLaunchKernelGGL(wprep_kernel, dim3(1024), dim3(256), 0, 0, dw, dwf, N, K);
    DArgs a = {};
    a.x = dx; a.wf = dwf; a.part = dpart; a.M = M; a.N = N; a.K = K; a.splits = splits; a.steps = steps; a.rotate = argc > 2 ? atoi(argv[2]) : 1;
    const int grid = (M / 128) * (N / 128) * splits;
    const size_t lds_bytes = 65536;
    auto k = dense_fwd_kernel<27, BDEPTH, ADEPTH>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const size_t total4 = (size_t)M * N / 4;
    auto launch = [&](hipStream_t st, const DArgs& aa) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds_bytes, st, aa);
        hipLaunchKernelGGL(fold_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, (const float4*)dpart, splits, total4, db, N, 1, (float4*)dy);
    };
    for (int i = 0; i < 3; ++i) launch(0, a);
    CK(hipDeviceSynchronize());
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int with_fold = 0; with_fold < 2; ++with_fold) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 20; ++i) {
            if (with_fold) launch(st, a);
            else hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds_bytes, st, a);
        }
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("dense forward %d x %d x %d, %d splits, grid %d, rotate %d%s: %.2f us per launch in a graph\n", M, N, K, splits, grid, a.rotate,
               with_fold ? " + fold" : "", ms / 100 * 1e3);
    }
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(wprep_kernel, dim3(1024), dim3(256), 0, st, dw, dwf, N, K);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("   weight prep (split + fragment order, %.1f MB in, %.1f MB out): %.2f us per launch\n", N * (double)K * 4e-6, N * (double)K * 6e-6, ms / 100 * 1e3);
    }
    {
        unsigned long long* dtr; CK(hipMalloc(&dtr, (size_t)grid * 8 * 32));
        DArgs at = a; at.trace = dtr;
        hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds_bytes, 0, at);
        std::vector<unsigned long long> tr((size_t)grid * 8 * 4);
        CK(hipMemcpy(tr.data(), dtr, (size_t)grid * 8 * 32, hipMemcpyDeviceToHost));
        double p0 = 0, p1 = 0, p2 = 0;
        for (int i = 0; i < grid * 8; ++i) { p0 += tr[4 * i + 1] - tr[4 * i]; p1 += tr[4 * i + 2] - tr[4 * i + 1]; p2 += tr[4 * i + 3] - tr[4 * i + 2]; }
        printf("   cycles per wave: prologue %.0f, loop %.0f (matrix-pipe time of its MFMAs: %d), epilogue %.0f\n", p0 / (grid * 8), p1 / (grid * 8),
               (steps + 1) / 2 * 36 * 32 * 2, p2 / (grid * 8));
    }
    launch(0, a);
    CK(hipDeviceSynchronize());
    std::vector<float> gy((size_t)M * N); std::vector<double> ry((size_t)M * N);
    CK(hipMemcpy(gy.data(), dy, gy.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ry.data(), dref, ry.size() * 8, hipMemcpyDeviceToHost));
    double num = 0, den = 0, worst = 0, big = 0;
    for (size_t i = 0; i < gy.size(); ++i) { const double d = gy[i] - ry[i]; num += d * d; den += ry[i] * ry[i]; if (fabs(d) > worst) worst = fabs(d); if (fabs(ry[i]) > big) big = fabs(ry[i]); }
    printf("   y: rms err / rms ref %.3g, max |err| %.3g (max |ref| %.3g): %s\n", sqrt(num / den), worst, big, sqrt(num / den) < 1e-6 ? "ok" : "FAIL");
    return 0;
}
