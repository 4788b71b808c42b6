// Prototype (round 4): weight gradient of conv 1 from the u8 observations, second image-stationary structure.
//   dw[n][c][ty][tx] = scale * sum over (image, oy, ox) of dy[image][oy][ox][n] * pixel[image][c][4 oy + ty][4 ox + tx]
// Why: the production kernel (wgrad_split_kernel<1,4,..>, 30.7 us at the PPO minibatch, 0.13 of the bf16 pipe) gathers the
// patch operand one dword per lane through the texture path -- PMC (tools/conv1_pmc.sh): 16.7 M L1 accesses per launch =
// 65 k per CU, i.e. the launch is bound by L1 ACCESSES, 48 % of the wave time parked at s_waitcnt.  Round 3's
// image-stationary prototype (conv1_wgrad_proto.hip) staged BOTH operands per image in LDS -- dy transposed and split
// into three bf16 planes, 118 KB: no room for a second image, staging and MFMAs alternate (31.4 us).  Here:
//   * only the image is staged -- as bf16 (exact), de-interleaved by column phase, S[c][y][x & 3][x >> 2] (a patch
//     fragment = one 16-byte LDS read + a funnel shift; conflict-free: see the kernel), 79 872 bytes per image: TWO
//     buffers, the next image's copy and conversion run under this image's MFMAs;
//   * dy never touches LDS: a fragment -- 8 consecutive output pixels of one filter -- is 8 dword loads per lane (128
//     consecutive bytes per half-wave), split into its three pieces in registers ONCE per reduction step because a wave
//     owns ALL eight tap tiles (128 accumulator registers), loaded two steps ahead;
//   * a wave keeps its 32 x 256 sums over all images of its workgroup; the waves' sums meet in LDS at the end (two rounds
//     of four tap tiles, fixed order): ONE partial per workgroup.
// MEASURED (MI355X, 512 images, one launch in a graph, fold not included; all variants correct, rms error 3.5e-7 of the
// float64 reference):
//   u8 image in LDS + in-loop conversion, dy one step ahead (first version)      37.3 us   steps of an image 24.5 k cycles
//     ... with the dy loads knocked out 27.5 us / 13.5 k; the LDS reads 36.1; the MFMAs 34.4; prologue 8.3 k, the
//     three-round reduction + store 18 k cycles
//   this file (bf16 image, dy two steps ahead, two-round reduction)              34-35.5 us   21-22 k (11.3 k without dy)
//   ... + one touch load per 128-byte line of the next image's dy (L2 prefetch)  no change
//   four waves of 512 registers, the next image's dy loaded a whole image ahead   54 us  (one wave per SIMD: every LDS /
//     barrier / issue stall is exposed)
// NOT adopted: what bounds it is the dy stream -- 32 KB in flight per CU (8 waves x 16 dword loads x 256 B) against a
// 2-4 k cycle path -- and 239 of 256 registers are taken (128 accumulators), so it cannot be deepened; without that
// stream the structure would run at 23 us, still behind what its MFMA time (7.2 k cycles per image) promises because the
// per-image barrier leaves two waves per SIMD nothing to overlap with.  usage: conv1_wgrad_v2 [images]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
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
__device__ __forceinline__ u32x2 bytes_to_bf16x4(unsigned v) {
    const float f0 = (float)(v & 0xffu), f1 = (float)((v >> 8) & 0xffu), f2 = (float)((v >> 16) & 0xffu), f3 = (float)(v >> 24);
    return u32x2{hi_pair(f0, f1), hi_pair(f2, f3)};
}

struct WgArgs {
    const unsigned char* obs;   // u8 [rows][C][H][W]
    const int* idx;             // row of image b, or null
    const float* dy;            // f32 [B][OH][OW][32]
    float* part;                // f32 [grid][32][C * 64]
    float* bias_part;           // f32 [grid][32] or null
    float scale;
    int n_img, C, H, W, OH, OW;
    unsigned long long* trace;
};

constexpr int NW = 8, NT = NW * 64, PX = 24, NTILE = 8;     // waves, threads, S row pitch (elements), tap tiles of 32 (C = 4)

// S (LDS, per image): the image as bf16 (exact: 0 .. 255), de-interleaved by column phase, S[c][y][x & 3][x >> 2], 24
// elements per (row, phase) (20 real + 4 that the padded output pixels read: they meet dy = 0).  An MFMA fragment of the
// patch operand -- tap (c, ty, tx), 8 consecutive output pixels of output row oy -- is 8 consecutive elements of
// S[c][4 oy + ty][tx & 3] from x0 + (tx >> 2): one 16-byte read (+ 4 bytes and a funnel shift for tx >= 4).  The phases of
// a row lie 48 bytes apart and rows 192: the 16 (ty & 3, phase) pairs of a half-wave fall into 16 different bank
// quartets -- no conflicts.  79 872 bytes per image: two buffers fill the CU's 160 KiB.
__global__ __launch_bounds__(NT) void conv1_wgrad_v2_kernel(const WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int C = a.C, H = a.H, W = a.W, OH = a.OH, OW = a.OW, K = C * 64;
    const int gpr = (OW + 7) / 8;                                   // 8-pixel groups per output row (19 -> 3, the last one padded)
    const int nq = OH * gpr, nsteps = (nq + 1) / 2;                 // groups per image; 16-pixel reduction steps
    const int row_b = 4 * PX * 2;                                   // bytes per image row in S (4 phases)
    const int s_bytes = C * H * row_b;
    const int s_pitch = s_bytes + 32;                               // (the last row's padded pixels read 4 bytes past it)
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if (a.trace) t0 = __builtin_readcyclecounter();
    // ---- this lane's taps: column l31 of tap tile jt is k = 32 jt + l31 = (c, ty, tx), c = jt >> 1, ty = 4 (jt & 1) + (l31 >> 3)
    const int ty0 = l31 >> 3, tx = l31 & 7;
    const unsigned sh = (tx >> 2) * 2;                              // bytes to funnel-shift by
    const unsigned base0 = (unsigned)(ty0 * row_b + (tx & 3) * (PX * 2));
    f32x16 acc[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    float bsum = 0.f;
    const int n_chunks = C * H * W / 16;
    constexpr int IMT = 5;                                          // 16-byte image chunks per thread (2080 / 512)
    u32x4 ri[IMT];
    auto issue = [&](int img) {
        const int row = a.idx ? a.idx[img] : img;
        const u32x4* src = reinterpret_cast<const u32x4*>(a.obs + (size_t)row * C * H * W);
#pragma unroll
        for (int i = 0; i < IMT; ++i) {
            const int t = tid + NT * i;
            ri[i] = src[t < n_chunks ? t : n_chunks - 1];
        }
    };
    auto stage = [&](char* sS) {
#pragma unroll
        for (int i = 0; i < IMT; ++i) {
            const int t = tid + NT * i;
            if (t < n_chunks) {
                const u32x4 v = ri[i];
                const int pr = t / (W / 16), xq = t - pr * (W / 16);        // (plane, row) and 16-pixel chunk of the row
                char* d = sS + pr * row_b + xq * 8;                         // elements 4 xq .. 4 xq + 3 of each phase
#pragma unroll
                for (int j = 0; j < 4; ++j) {                               // phase j: bytes j, j + 4, j + 8, j + 12 of the chunk
                    const unsigned b4 = __builtin_amdgcn_perm(v.y, v.x, 0x0c0c0400u + 0x0101u * j) |
                                        (__builtin_amdgcn_perm(v.w, v.z, 0x0c0c0400u + 0x0101u * j) << 16);
                    *reinterpret_cast<u32x2*>(d + j * (PX * 2)) = bytes_to_bf16x4(b4);
                }
            }
        }
    };
    // dy fragment of step s of image img: 8 consecutive output pixels (group q = 2 s + half) of filter l31
    auto load_a = [&](int img, int s, float (&r)[8]) {
        const int q = 2 * s + half;
        const int qq = q < nq ? q : nq - 1;
        const int oy = qq / gpr, x0 = 8 * (qq - gpr * oy);
        const float* p = a.dy + ((size_t)img * OH * OW + (size_t)oy * OW) * 32 + l31;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ox = x0 + i;
            const float v = p[(ox < OW ? ox : OW - 1) * 32];
            r[i] = (q < nq && ox < OW) ? v : 0.f;
        }
    };
    if (tid < 4) *reinterpret_cast<u32x4*>(lds + s_bytes + (tid & 1) * 16 + (tid >> 1) * s_pitch) = u32x4{0, 0, 0, 0};   // the slack
    // elements W / 4 .. PX - 1 of every (row, phase) are never staged and only met by dy = 0 -- but 0 x NaN is NaN: zero once
    for (int i = tid; i < 2 * C * H * 4; i += NT) {
        char* d = lds + (i >= C * H * 4 ? s_pitch : 0) + (i % (C * H * 4)) * (PX * 2) + (W / 4) * 2;
        for (int e = 0; e < (PX - W / 4) * 2; e += 4) *reinterpret_cast<unsigned*>(d + e) = 0u;
    }
    issue(blockIdx.x);
    float ar0[8], ar1[8];                                           // the dy values of this wave's next two steps
    load_a(blockIdx.x, wave, ar0);
    load_a(blockIdx.x, wave + NW, ar1);
    stage(lds);
    int buf = 0;
    // one reduction step: split the dy fragment, then (the values are consumed) refill the registers with the step two
    // ahead -- of this image, or the first / second step of the next one -- and run the eight tap tiles
    auto step = [&](const char* sS, int s, float (&ar)[8], int img_next, int s_next) {
        u32x4 fa[3];
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            unsigned h, m, l;
            split_pair(ar[2 * pr], ar[2 * pr + 1], h, m, l);
            fa[0][pr] = h; fa[1][pr] = m; fa[2][pr] = l;
            bsum += ar[2 * pr]; bsum += ar[2 * pr + 1];
        }
        if (img_next < a.n_img) load_a(img_next, s_next, ar);
        const int q = 2 * s + half;
        const int qq = q < nq ? q : nq - 1;
        const int oy = qq / gpr, x0 = 8 * (qq - gpr * oy);
        const char* p0 = sS + base0 + (unsigned)(4 * oy * row_b + 2 * x0);
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
            const char* p = p0 + ((t >> 1) * H + 4 * (t & 1)) * row_b;
            const u32x4 d = *reinterpret_cast<const u32x4*>(p);
            const unsigned d4 = *reinterpret_cast<const unsigned*>(p + 16);
            const u32x4 fb = u32x4{__builtin_amdgcn_alignbyte(d.y, d.x, sh), __builtin_amdgcn_alignbyte(d.z, d.y, sh),
                                   __builtin_amdgcn_alignbyte(d.w, d.z, sh), __builtin_amdgcn_alignbyte(d4, d.w, sh)};
#pragma unroll
            for (int pl = 2; pl >= 0; --pl) acc[t] = mfma_bf16(fa[pl], fb, acc[t]);
        }
    };
    for (int img = blockIdx.x; img < a.n_img; img += gridDim.x) {
        const int nxt = img + (int)gridDim.x;
        if (nxt < a.n_img) issue(nxt);
        __syncthreads();                                            // this image's bytes are in LDS; the other buffer is free
        if (a.trace && img == (int)blockIdx.x) t1 = __builtin_readcyclecounter();
        const char* sS = lds + buf * s_pitch;
        // steps wave, wave + NW, ... in pairs (a step past the end is skipped: its registers already hold the next image's)
        for (int s = wave; s < nsteps; s += 2 * NW) {
            const int sa = s + 2 * NW, sb = s + 3 * NW;
            step(sS, s, ar0, sa < nsteps ? img : nxt, sa < nsteps ? sa : wave);
            if (s + NW < nsteps) step(sS, s + NW, ar1, sb < nsteps ? img : nxt, sb < nsteps ? sb : wave + NW);
        }
        if (a.trace && img == (int)blockIdx.x) t2 = __builtin_readcyclecounter();
        if (nxt < a.n_img) stage(lds + (buf ^ 1) * s_pitch);        // (every wave left that buffer before the last barrier)
        buf ^= 1;
    }
    // ---- the waves' sums meet through LDS: in two rounds of four tap tiles every wave leaves its sums, wave w (< 4) adds
    // the eight contributions to tile 4 round + w in wave order and writes that tile of the workgroup's partial
    float* red = reinterpret_cast<float*>(lds);                     // [NW][4][16][64] floats = 128 KB
    float* out = a.part + (size_t)blockIdx.x * 32 * K;
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) red[((wave * 4 + t) * 16 + v) * 64 + lane] = acc[4 * round + t][v];
        __syncthreads();
        if (wave < 4) {
            const int k = 32 * (4 * round + wave) + l31;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float sum = red[((0 * 4 + wave) * 16 + v) * 64 + lane];
#pragma unroll
                for (int w = 1; w < NW; ++w) sum += red[((w * 4 + wave) * 16 + v) * 64 + lane];
                const int n = (v & 3) + 8 * (v >> 2) + 4 * half;
                if (k < K) out[n * K + k] = sum * a.scale;
            }
        }
    }
    // bias gradient: every lane summed the dy values of filter l31 it loaded; halves, then waves, in a fixed order
    __syncthreads();
    float* bred = reinterpret_cast<float*>(lds);
    bsum += __shfl_xor(bsum, 32, 64);
    if (lane < 32) bred[wave * 32 + lane] = bsum;
    __syncthreads();
    if (a.bias_part && tid < 32) {
        float s = 0.f;
        for (int w = 0; w < NW; ++w) s += bred[w * 32 + tid];
        a.bias_part[(size_t)blockIdx.x * 32 + tid] = s;
    }
    if (a.trace && lane == 0) {
        unsigned long long* t = a.trace + (blockIdx.x * NW + wave) * 4;
        t[0] = t0; t[1] = t1; t[2] = t2; t[3] = __builtin_readcyclecounter();
    }
}

__global__ void fold_kernel(const float* part, int splits, int total, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float s = 0;
    for (int z = 0; z < splits; ++z) s += part[(size_t)z * total + i];
    out[i] = s;
}

__global__ void ref_kernel(const unsigned char* obs, const int* idx, const float* dy, double* dw, double* db, int n_img, int C, int H, int W,
                           int OH, int OW, float scale) {
    const int K = C * 64, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 32 * K + 32) return;
    if (i >= 32 * K) {
        const int n = i - 32 * K; double s = 0;
        for (size_t r = 0; r < (size_t)n_img * OH * OW; ++r) s += dy[r * 32 + n];
        db[n] = s; return;
    }
    const int n = i / K, k = i % K, c = k >> 6, ty = (k >> 3) & 7, tx = k & 7;
    double s = 0;
    for (int b = 0; b < n_img; ++b) {
        const unsigned char* im = obs + (size_t)(idx ? idx[b] : b) * C * H * W;
        for (int oy = 0; oy < OH; ++oy)
            for (int ox = 0; ox < OW; ++ox)
                s += (double)dy[((size_t)(b * OH + oy) * OW + ox) * 32 + n] * (double)im[(c * H + 4 * oy + ty) * W + 4 * ox + tx];
    }
    dw[i] = s * (double)scale;
}

int main(int argc, char** argv) {
    const int n_img = argc > 1 ? atoi(argv[1]) : 512;
    const int C = 4, H = 104, W = 80, OH = 25, OW = 19, n_rows = n_img + 77, K = C * 64;
    const size_t nobs = (size_t)n_rows * C * H * W, ndy = (size_t)n_img * OH * OW * 32;
    std::vector<unsigned char> ho(nobs);
    std::vector<float> hdy(ndy);
    std::vector<int> hidx(n_img);
    srand(5);
    for (auto& v : ho) v = rand() & 255;
    for (auto& v : hdy) v = (rand() & 1) ? 0.f : ((rand() / (float)RAND_MAX) - 0.5f) * 0.01f;
    for (int i = 0; i < n_img; ++i) hidx[i] = (i * 7919) % n_rows;
    const int grid = n_img < 256 ? n_img : 256;
    unsigned char* dobs; float *ddy, *dpart, *dbpart, *ddw, *ddb; double *drw, *drb; int* didx;
    CK(hipMalloc(&dobs, nobs)); CK(hipMalloc(&ddy, ndy * 4)); CK(hipMalloc(&dpart, (size_t)grid * 32 * K * 4));
    CK(hipMalloc(&dbpart, grid * 128)); CK(hipMalloc(&ddw, 32 * K * 4)); CK(hipMalloc(&ddb, 128));
    CK(hipMalloc(&drw, 32 * K * 8)); CK(hipMalloc(&drb, 256)); CK(hipMalloc(&didx, n_img * 4));
    CK(hipMemcpy(dobs, ho.data(), nobs, hipMemcpyHostToDevice));
    CK(hipMemcpy(ddy, hdy.data(), ndy * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(didx, hidx.data(), n_img * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ref_kernel, dim3((32 * K + 32 + 63) / 64), dim3(64), 0, 0, dobs, didx, ddy, drw, drb, n_img, C, H, W, OH, OW, 1.f / 255.f);
    WgArgs a = {};
    a.obs = dobs; a.idx = didx; a.dy = ddy; a.part = dpart; a.bias_part = dbpart; a.scale = 1.f / 255.f;
    a.n_img = n_img; a.C = C; a.H = H; a.W = W; a.OH = OH; a.OW = OW;
    const size_t s_pitch = (size_t)C * H * 4 * PX * 2 + 32;
    const size_t red_bytes = (size_t)NW * 4 * 16 * 64 * 4;
    const size_t lds_bytes = 2 * s_pitch > red_bytes ? 2 * s_pitch : red_bytes;
    auto k = conv1_wgrad_v2_kernel;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    auto launch = [&](hipStream_t st, const WgArgs& aa) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds_bytes, st, aa);
        hipLaunchKernelGGL(fold_kernel, dim3((32 * K + 255) / 256), dim3(256), 0, st, dpart, grid, 32 * K, ddw);
        hipLaunchKernelGGL(fold_kernel, dim3(1), dim3(32), 0, st, dbpart, grid, 32, ddb);
    };
    for (int i = 0; i < 3; ++i) launch(0, a);
    CK(hipDeviceSynchronize());
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds_bytes, st, a);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("conv 1 weight gradient from u8, v2 (dy from global, image double-buffered): %d images, grid %d, lds %zu: %.2f us per launch in a graph (fold not included)\n",
           n_img, grid, lds_bytes, ms / 100 * 1e3);
    {
        unsigned long long* dtr; CK(hipMalloc(&dtr, (size_t)grid * NW * 32));
        WgArgs at = a; at.trace = dtr;
        hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds_bytes, 0, at);
        std::vector<unsigned long long> tr((size_t)grid * NW * 4);
        CK(hipMemcpy(tr.data(), dtr, (size_t)grid * NW * 32, hipMemcpyDeviceToHost));
        double p0 = 0, p1 = 0, p2 = 0;
        for (int i = 0; i < grid * NW; ++i) { p0 += tr[4 * i + 1] - tr[4 * i]; p1 += tr[4 * i + 2] - tr[4 * i + 1]; p2 += tr[4 * i + 3] - tr[4 * i]; }
        printf("   cycles per wave: until the first image is staged %.0f, its MFMA steps %.0f, whole %.0f\n", p0 / (grid * NW), p1 / (grid * NW), p2 / (grid * NW));
    }
    launch(0, a);
    CK(hipDeviceSynchronize());
    std::vector<float> gw(32 * K), gb(32); std::vector<double> rw(32 * K), rb(32);
    CK(hipMemcpy(gw.data(), ddw, 32 * K * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gb.data(), ddb, 128, hipMemcpyDeviceToHost));
    CK(hipMemcpy(rw.data(), drw, 32 * K * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(rb.data(), drb, 256, hipMemcpyDeviceToHost));
    double num = 0, den = 0, worst = 0, big = 0;
    for (int i = 0; i < 32 * K; ++i) { const double d = gw[i] - rw[i]; num += d * d; den += rw[i] * rw[i]; if (fabs(d) > worst) worst = fabs(d); if (fabs(rw[i]) > big) big = fabs(rw[i]); }
    double bw = 0, bb = 0;
    for (int i = 0; i < 32; ++i) { if (fabs(gb[i] - rb[i]) > bw) bw = fabs(gb[i] - rb[i]); if (fabs(rb[i]) > bb) bb = fabs(rb[i]); }
    printf("   dw: rms err / rms ref %.3g, max |err| %.3g (max |ref| %.3g): %s;  db: max |err| %.3g (max |ref| %.3g): %s\n", sqrt(num / den), worst, big,
           sqrt(num / den) < 2e-6 ? "ok" : "FAIL", bw, bb, bw <= 1e-5 * (1 + bb) ? "ok" : "FAIL");
    return 0;
}
