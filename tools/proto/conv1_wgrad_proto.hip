// Prototype (round 3): weight gradient of conv 1 from the u8 observations, "image-stationary".
//   dw[n][c][ty][tx] = scale * sum over (image, oy, ox) of dy[image][oy][ox][n] * pixel[image][c][4 oy + ty][4 ox + tx]
// A persistent 1024-thread workgroup per CU keeps the 32 x 256 result in its waves' accumulators over all of its images
// (wave w: the 32 columns k = 32 (w & 7) .. + 31, half (w >> 3) of an image's reduction steps) and leaves ONE partial
// per workgroup for the fold.  Per image, in LDS:
//   T  dy transposed and split: three bf16 planes [32 filters][r], r = 24 oy + ox (an output row padded from 19 to 24
//      pixels, zeros in the padding), so that an MFMA fragment -- 8 consecutive r of one filter -- is one 16-byte read;
//   S  the u8 image de-interleaved by column phase, S[x & 3][c][y][x >> 2]: for a fixed filter tap (c, ty, tx) eight
//      consecutive output pixels of a row read EIGHT CONSECUTIVE BYTES of S[tx & 3] (stride 4 in x became stride 1) --
//      the other fragment, converted to bf16 (exact) between the LDS read and the MFMA.
// The kernel it would replace gathers one dword per lane and instruction from global memory (texture-addresser bound).
// Checks against a float64 reference and times it.  usage: conv1_wgrad_proto [images]
// MEASURED (MI355X, 512 images of 4 x 104 x 80, one launch in a graph, fold not included): 30.8 us with the loads of an
// image issued where they are used, 32.3 us with the next image's loads held in registers over the MFMA steps (96 VGPRs);
// per wave and image ~10 k cycles of staging (split + 36 LDS dword writes per thread) and ~9.6 k cycles for its 57 MFMAs
// (4 waves per SIMD: VALU byte->bf16 conversion ~= MFMA time).  The production gather kernel + fold takes ~29.8 us: NOT
// adopted; results are correct (rms error 3.7e-7 of the float64 reference).
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

constexpr int NW = 16, NT = NW * 64, RP = 24, TP = 616, PX = 24;    // waves, threads, padded output-row length, T pitch, S pitch

__global__ __launch_bounds__(NT) void conv1_wgrad_img_kernel(const WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int C = a.C, H = a.H, W = a.W, OH = a.OH, OW = a.OW, K = C * 64;
    const int nq = (OH * 3 + 1) / 2 * 2, nsteps = nq / 2;          // 8-pixel chunks of the padded reduction space, 16-r steps
    const int t_bytes = 3 * 32 * TP * 2;
    char* const sT = lds;                                           // [3][32][TP] bf16
    char* const sS = lds + t_bytes;                                 // [4][C][H][PX] u8
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if (a.trace) t0 = __builtin_readcyclecounter();
    for (int i = tid; i < t_bytes / 16; i += NT) *reinterpret_cast<u32x4*>(sT + i * 16) = u32x4{0, 0, 0, 0};
    const int w4 = W / 4, s_bytes = 4 * C * H * PX;
    for (int i = tid; i < (s_bytes + 512) / 16; i += NT) *reinterpret_cast<u32x4*>(sS + i * 16) = u32x4{0, 0, 0, 0};
    // ---- this lane's column of the result: k = 32 (wave & 7) + l31 = (c, ty, tx)
    const int jt = wave & 7, g = wave >> 3;
    const int k = jt * 32 + l31;
    const int kc = k >> 6, kty = (k >> 3) & 7, ktx = k & 7;
    const unsigned dxs = ktx >> 2;
    const unsigned sbase = (unsigned)((((ktx & 3) * C + kc) * H + kty) * PX);      // + 4 oy * PX + x0 (+ dxs, funnel-shifted)
    const bool col_ok = k < K;
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    const int s_lo = g ? (nsteps + 1) / 2 : 0, s_hi = g ? nsteps : (nsteps + 1) / 2;
    const int pairs_x = (OW + 1) / 2, n_tasks = OH * pairs_x * 8, n_chunks = C * H * W / 16;
    // staging registers: the NEXT image's dy pixel pairs and image bytes are loaded while this image's steps run
    constexpr int DYT = 2, IMT = 3;                                 // tasks per thread (2000 / 1024, 2080 / 1024)
    float4 r0[DYT], r1[DYT];
    u32x4 ri[IMT];
    auto issue = [&](int img) {
        const float* dyi = a.dy + (size_t)img * OH * OW * 32;
#pragma unroll
        for (int i = 0; i < DYT; ++i) {
            const int t = tid + NT * i;
            r0[i] = r1[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < n_tasks) {
                const int c4 = t & 7, pr = t >> 3, oy = pr / pairs_x, ox = 2 * (pr - oy * pairs_x);
                r0[i] = *reinterpret_cast<const float4*>(dyi + (oy * OW + ox) * 32 + c4 * 4);
                if (ox + 1 < OW) r1[i] = *reinterpret_cast<const float4*>(dyi + (oy * OW + ox + 1) * 32 + c4 * 4);
            }
        }
        const int row = a.idx ? a.idx[img] : img;
        const u32x4* src = reinterpret_cast<const u32x4*>(a.obs + (size_t)row * C * H * W);
#pragma unroll
        for (int i = 0; i < IMT; ++i)
            if (tid + NT * i < n_chunks) ri[i] = src[tid + NT * i];
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < DYT; ++i) {
            const int t = tid + NT * i;
            if (t < n_tasks) {
                const int c4 = t & 7, pr = t >> 3, oy = pr / pairs_x, ox = 2 * (pr - oy * pairs_x);
                const float4 v0 = r0[i], v1 = r1[i];
                bsum.x += v0.x + v1.x; bsum.y += v0.y + v1.y; bsum.z += v0.z + v1.z; bsum.w += v0.w + v1.w;
                const float e0[4] = {v0.x, v0.y, v0.z, v0.w}, e1[4] = {v1.x, v1.y, v1.z, v1.w};
                char* d = sT + (c4 * 4) * (TP * 2) + (oy * RP + ox) * 2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned h, m, l;
                    split_pair(e0[e], e1[e], h, m, l);
                    *reinterpret_cast<unsigned*>(d + e * (TP * 2)) = h;
                    *reinterpret_cast<unsigned*>(d + e * (TP * 2) + 32 * TP * 2) = m;
                    *reinterpret_cast<unsigned*>(d + e * (TP * 2) + 64 * TP * 2) = l;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < IMT; ++i) {
            const int t = tid + NT * i;
            if (t < n_chunks) {
                const u32x4 v = ri[i];
                const int pr = t / (W / 16), xq = t - pr * (W / 16);
                char* d = sS + pr * PX + xq * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned lo = __builtin_amdgcn_perm(v.y, v.x, 0x0c0c0400u + 0x0101u * j);
                    const unsigned hi = __builtin_amdgcn_perm(v.w, v.z, 0x0c0c0400u + 0x0101u * j);
                    *reinterpret_cast<unsigned*>(d + j * (C * H * PX)) = lo | (hi << 16);
                }
            }
        }
    };
    issue(blockIdx.x);
    for (int img = blockIdx.x; img < a.n_img; img += gridDim.x) {
        __syncthreads();                                            // the previous image's fragments are read
        stage();
        if (img + (int)gridDim.x < a.n_img) issue(img + gridDim.x);
        __syncthreads();
        if (a.trace && img == (int)blockIdx.x) t1 = __builtin_readcyclecounter();
        // ---- this wave's steps of the image
        const char* ta = sT + l31 * (TP * 2) + half * 16;
#pragma unroll 2
        for (int s = s_lo; s < s_hi; ++s) {
            const int q = 2 * s + half;
            const int oy = (q * 171) >> 9, x0 = 8 * (q - 3 * oy);   // q / 3 for q < 512
            const char* p = sS + sbase + (unsigned)(4 * oy * PX + x0);
            const u32x2 d01 = *reinterpret_cast<const u32x2*>(p);
            const unsigned d2 = *reinterpret_cast<const unsigned*>(p + 8);
            const unsigned w0 = __builtin_amdgcn_alignbyte(d01.y, d01.x, dxs), w1 = __builtin_amdgcn_alignbyte(d2, d01.y, dxs);
            const u32x2 b0 = bytes_to_bf16x4(w0), b1 = bytes_to_bf16x4(w1);
            const u32x4 fb = u32x4{b0.x, b0.y, b1.x, b1.y};
            u32x4 fa[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fa[pl] = *reinterpret_cast<const u32x4*>(ta + pl * (32 * TP * 2) + s * 32);
#pragma unroll
            for (int pl = 2; pl >= 0; --pl) acc = mfma_bf16(fa[pl], fb, acc);
        }
        if (a.trace && img == (int)blockIdx.x) t2 = __builtin_readcyclecounter();
    }
    // ---- the two halves of the reduction meet through LDS; the workgroup's partial leaves
    __syncthreads();
    float* red = reinterpret_cast<float*>(sT);
    if (g == 1) {
#pragma unroll
        for (int v = 0; v < 16; ++v) red[(jt * 16 + v) * 64 + lane] = acc[v];
    }
    float4* bred = reinterpret_cast<float4*>(sT + 8 * 16 * 64 * 4);
    bred[tid] = bsum;
    __syncthreads();
    if (g == 0 && col_ok) {
        float* out = a.part + (size_t)blockIdx.x * 32 * K;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int n = (v & 3) + 8 * (v >> 2) + 4 * half;
            out[n * K + k] = (acc[v] + red[(jt * 16 + v) * 64 + lane]) * a.scale;
        }
    }
    if (a.bias_part && tid < 8) {                                   // channels 4 tid .. + 3: threads tid, tid + 8, ... in order
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = tid; i < NT; i += 8) { const float4 v = bred[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        *reinterpret_cast<float4*>(a.bias_part + (size_t)blockIdx.x * 32 + tid * 4) = s;
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
    const size_t lds_bytes = (size_t)3 * 32 * TP * 2 + (size_t)4 * C * H * PX + 512;
    auto k = conv1_wgrad_img_kernel;
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
    printf("conv 1 weight gradient from u8, image-stationary: %d images, grid %d, lds %zu: %.2f us per launch in a graph (fold not included)\n",
           n_img, grid, lds_bytes, ms / 100 * 1e3);
    {
        unsigned long long* dtr; CK(hipMalloc(&dtr, (size_t)grid * NW * 32));
        WgArgs at = a; at.trace = dtr;
        hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds_bytes, 0, at);
        std::vector<unsigned long long> tr((size_t)grid * NW * 4);
        CK(hipMemcpy(tr.data(), dtr, (size_t)grid * NW * 32, hipMemcpyDeviceToHost));
        double p0 = 0, p1 = 0, p2 = 0;
        for (int i = 0; i < grid * NW; ++i) { p0 += tr[4 * i + 1] - tr[4 * i]; p1 += tr[4 * i + 2] - tr[4 * i + 1]; p2 += tr[4 * i + 3] - tr[4 * i]; }
        printf("   cycles per wave: zero + stage the first image %.0f, its MFMA steps %.0f, whole %.0f\n", p0 / (grid * NW), p1 / (grid * NW), p2 / (grid * NW));
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
