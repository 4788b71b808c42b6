// Prototype (round 3): conv 1 straight from the u8 observations, "image-stationary".
//   * a persistent 512-thread workgroup per CU; the layer's weights, split exactly into three bf16 planes in MFMA-fragment
//     order, live in LDS for the workgroup's whole life (48 KB at 32 filters x 256 taps);
//   * per image: the planar u8 image (33 KB) arrives with coalesced 16-byte loads, is converted ONCE per pixel to bf16
//     (0 .. 255 is exact) and stored in LDS (66 KB); every MFMA A fragment -- eight consecutive pixels of one filter row
//     -- is then one 16-byte LDS read: no address gymnastics, no per-tap conversion, no global gathers;
//   * a wave owns two 32-pixel row tiles and walks the 16 k-steps with 6 MFMAs (2 tiles x 3 weight planes) per step.
// The current kernel gathers 4-byte pieces per lane from global memory (64 cache lines per load instruction).
// Checks against a float64 reference and times it.  usage: conv1_img_proto [images]
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
// four packed bytes -> four bf16 (two dwords), exact
__device__ __forceinline__ u32x2 bytes_to_bf16x4(unsigned v) {
    const float f0 = (float)(v & 0xffu), f1 = (float)((v >> 8) & 0xffu), f2 = (float)((v >> 16) & 0xffu), f3 = (float)(v >> 24);
    return u32x2{hi_pair(f0, f1), hi_pair(f2, f3)};
}

struct Conv1Args {
    const unsigned char* obs;   // u8 [rows][C][H][W]
    const int* idx;             // row of image b, or null
    const float* w;             // f32 [32][C][8][8]
    const float* bias;          // f32 [32] or null
    float* y;                   // f32 [B][OH][OW][32]
    float scale;
    int n_img, C, H, W, OH, OW, stride, relu;
    unsigned long long* trace;
};

// V2: the image stays u8 in LDS (two buffers: the next image's bytes land while this one is computed; one barrier per
// image), a fragment's eight pixels are converted between the LDS read and the MFMA (12 vector instructions per fragment,
// 4 per MFMA: the layer needs three products per multiply, the matrix pipe has room).
template <int TPW, int C, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void conv1_img_kernel(const Conv1Args a) {
    constexpr int NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int N = 32, KH = 8, KW = 8, K = C * KH * KW, NSTEPS = K / 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int H = a.H, W = a.W, npix = C * H * W;
    const int img_bytes = (npix + 15) & ~15;
    char* const sW = lds;                                   // [NSTEPS][3 planes][64 lanes] 16-byte fragments
    char* const sI = lds + NSTEPS * 3 * 1024;               // two u8 images [C][H][W]
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (a.trace) t0 = __builtin_readcyclecounter();
    constexpr int MAXLD = (40960 / 16 + NT - 1) / NT;       // 16-byte loads per thread that cover an image (<= 40 KB)
    const int n16 = npix / 16;
    u32x4 ireg[MAXLD];
    auto img_issue = [&](int img) {
        const int row = a.idx ? a.idx[img] : img;
        const u32x4* src = reinterpret_cast<const u32x4*>(a.obs + (size_t)row * npix);
#pragma unroll
        for (int i = 0; i < MAXLD; ++i)
            if (tid + NT * i < n16) ireg[i] = src[tid + NT * i];
    };
    auto img_store = [&](int buf) {
        char* d = sI + buf * img_bytes;
#pragma unroll
        for (int i = 0; i < MAXLD; ++i)
            if (tid + NT * i < n16) *reinterpret_cast<u32x4*>(d + (tid + NT * i) * 16) = ireg[i];
    };
    img_issue(blockIdx.x);
    for (int f = tid; f < NSTEPS * 64; f += NT) {          // weights: split once per workgroup into fragment order
        const int s = f >> 6, fl = f & 63;
        const float* src = a.w + (size_t)(fl & 31) * K + s * 16 + (fl >> 5) * 8;
        const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
        unsigned h[4], m[4], l[4];
        split_pair(v0.x, v0.y, h[0], m[0], l[0]);
        split_pair(v0.z, v0.w, h[1], m[1], l[1]);
        split_pair(v1.x, v1.y, h[2], m[2], l[2]);
        split_pair(v1.z, v1.w, h[3], m[3], l[3]);
        char* d = sW + s * 3072 + fl * 16;
        *reinterpret_cast<u32x4*>(d) = u32x4{h[0], h[1], h[2], h[3]};
        *reinterpret_cast<u32x4*>(d + 1024) = u32x4{m[0], m[1], m[2], m[3]};
        *reinterpret_cast<u32x4*>(d + 2048) = u32x4{l[0], l[1], l[2], l[3]};
    }
    const int rows = a.OH * a.OW, tiles = (rows + 31) / 32;
    float4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias) bq[q] = *reinterpret_cast<const float4*>(a.bias + 8 * q + 4 * half);
    }
    img_store(0);
    if (blockIdx.x + gridDim.x < a.n_img) img_issue(blockIdx.x + gridDim.x);
    __syncthreads();
    if (a.trace) t1 = __builtin_readcyclecounter();
    int buf = 0;
    for (int img = blockIdx.x; img < a.n_img; img += gridDim.x, buf ^= 1) {
        const char* im = sI + buf * img_bytes;
        for (int tp = wave * TPW; tp < tiles; tp += NW * TPW) {
            unsigned off[TPW];
            int m[TPW];
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                m[i] = (tp + i) * 32 + l31;
                const int mm = m[i] < rows ? m[i] : 0;
                const int oy = mm / a.OW, ox = mm - oy * a.OW;
                off[i] = (unsigned)((oy * a.stride + half) * W + ox * a.stride);             // filter row `half` of a step
            }
            f32x16 acc[TPW];
#pragma unroll
            for (int i = 0; i < TPW; ++i)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
            const char* wl = sW + lane * 16;
#pragma unroll 2
            for (int s = 0; s < NSTEPS; ++s) {              // step s = plane s / 4, filter rows 2 (s % 4) + half
                const unsigned po = (unsigned)(((s / (KH / 2)) * H + 2 * (s % (KH / 2))) * W);
                u32x4 fb[3], fa[TPW];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) fb[pl] = *reinterpret_cast<const u32x4*>(wl + s * 3072 + pl * 1024);
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    const unsigned* q = reinterpret_cast<const unsigned*>(im + po + off[i]);
                    const u32x2 lo = bytes_to_bf16x4(q[0]), hi = bytes_to_bf16x4(q[1]);
                    fa[i] = u32x4{lo.x, lo.y, hi.x, hi.y};
                }
#pragma unroll
                for (int pl = 2; pl >= 0; --pl)
#pragma unroll
                    for (int i = 0; i < TPW; ++i) acc[i] = mfma_bf16(fb[pl], fa[i], acc[i]);
            }
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                if (tp + i < tiles && m[i] < rows) {
                    float* dst = a.y + ((size_t)img * rows + m[i]) * N + 4 * half;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float4 v = make_float4(acc[i][4 * q] * a.scale + bq[q].x, acc[i][4 * q + 1] * a.scale + bq[q].y,
                                               acc[i][4 * q + 2] * a.scale + bq[q].z, acc[i][4 * q + 3] * a.scale + bq[q].w);
                        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        *reinterpret_cast<float4*>(dst + 8 * q) = v;
                    }
                }
            }
        }
        // the next image's bytes (loaded while this one was computed) -> the other buffer, which every wave left at the
        // last barrier; the image after that starts its way from memory
        if (img + (int)gridDim.x < a.n_img) {
            img_store(buf ^ 1);
            if (img + 2 * (int)gridDim.x < a.n_img) img_issue(img + 2 * gridDim.x);
        }
        if (a.trace && img == (int)blockIdx.x) t2 = __builtin_readcyclecounter();
        __syncthreads();
    }
    if (a.trace && lane == 0) {
        unsigned long long* t = a.trace + (blockIdx.x * NW + wave) * 4;
        t3 = __builtin_readcyclecounter();
        t[0] = t0; t[1] = t1; t[2] = t2; t[3] = t3;
    }
}

__global__ void ref_kernel(const unsigned char* obs, const int* idx, const float* w, const float* bias, double* y, int n_img,
                           int C, int H, int W, int OH, int OW, int stride, float scale) {
    const size_t total = (size_t)n_img * OH * OW * 32;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = i % 32; size_t t = i / 32;
        const int ox = t % OW; t /= OW;
        const int oy = t % OH; const int b = t / OH;
        const unsigned char* im = obs + (size_t)(idx ? idx[b] : b) * C * H * W;
        double s = 0;
        for (int c = 0; c < C; ++c)
            for (int ty = 0; ty < 8; ++ty)
                for (int tx = 0; tx < 8; ++tx)
                    s += (double)im[(c * H + oy * stride + ty) * W + ox * stride + tx] * (double)w[((n * C + c) * 8 + ty) * 8 + tx];
        s = s * (double)scale + bias[n];
        y[i] = s > 0 ? s : 0;
    }
}

int main(int argc, char** argv) {
    const int n_img = argc > 1 ? atoi(argv[1]) : 512;
    const int C = 4, H = 104, W = 80, stride = 4, OH = 25, OW = 19, n_rows = n_img + 77;
    const size_t nobs = (size_t)n_rows * C * H * W, ny = (size_t)n_img * OH * OW * 32, nw = 32 * C * 64;
    std::vector<unsigned char> ho(nobs);
    std::vector<float> hw(nw), hb(32);
    std::vector<int> hidx(n_img);
    srand(3);
    for (auto& v : ho) v = rand() & 255;
    for (auto& v : hw) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.2f;
    for (auto& v : hb) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.1f;
    for (int i = 0; i < n_img; ++i) hidx[i] = (i * 7919) % n_rows;
    unsigned char* dobs; float *dw, *db, *dy; double* dref; int* didx;
    CK(hipMalloc(&dobs, nobs)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&db, 128)); CK(hipMalloc(&dy, ny * 4));
    CK(hipMalloc(&dref, ny * 8)); CK(hipMalloc(&didx, n_img * 4));
    CK(hipMemcpy(dobs, ho.data(), nobs, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), 128, hipMemcpyHostToDevice));
    CK(hipMemcpy(didx, hidx.data(), n_img * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ref_kernel, dim3(2048), dim3(256), 0, 0, dobs, didx, dw, db, dref, n_img, C, H, W, OH, OW, stride, 1.f / 255.f);
    CK(hipDeviceSynchronize());
    Conv1Args a = {};
    a.obs = dobs; a.idx = didx; a.w = dw; a.bias = db; a.y = dy; a.scale = 1.f / 255.f;
    a.n_img = n_img; a.C = C; a.H = H; a.W = W; a.OH = OH; a.OW = OW; a.stride = stride; a.relu = 1;
    const size_t lds_bytes = (size_t)(C * 64 / 16) * 3 * 1024 + 2 * (size_t)C * H * W;
    const int grid = n_img < 256 ? n_img : 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto bench = [&](auto k, int nw, const char* what) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        CK(hipMemset(dy, 0xff, ny * 4));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(nw * 64), lds_bytes, 0, a);
        CK(hipDeviceSynchronize());
        // 20 launches inside one hipGraph, like the learner runs them
        hipStream_t st; CK(hipStreamCreate(&st));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(nw * 64), lds_bytes, st, a);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms / 100 * 1e3;
        printf("%-40s %d images, grid %d, lds %zu: %.2f us per launch in a graph (%.1f MB in + out -> %.2f TB/s)\n", what, n_img, grid,
               lds_bytes, us, (n_img * 33280.0 + ny * 4.0) * 1e-6, (n_img * 33280.0 + ny * 4.0) / (us * 1e-6) * 1e-12);
        unsigned long long* dtr; CK(hipMalloc(&dtr, (size_t)grid * nw * 32));
        Conv1Args at = a; at.trace = dtr;
        hipLaunchKernelGGL(k, dim3(grid), dim3(nw * 64), lds_bytes, 0, at);
        std::vector<unsigned long long> tr((size_t)grid * nw * 4);
        CK(hipMemcpy(tr.data(), dtr, (size_t)grid * nw * 32, hipMemcpyDeviceToHost));
        double p0 = 0, p1 = 0, p2 = 0;
        for (int i = 0; i < grid * nw; ++i) { p0 += tr[4 * i + 1] - tr[4 * i]; p1 += tr[4 * i + 2] - tr[4 * i + 1]; p2 += tr[4 * i + 3] - tr[4 * i]; }
        printf("   cycles per wave: prologue (weights + first image) %.0f, first image's tiles %.0f, whole %.0f\n", p0 / (grid * nw), p1 / (grid * nw), p2 / (grid * nw));
        CK(hipFree(dtr));
    };
    bench(conv1_img_kernel<2, 4, 8>, 8, "8 waves x 2 tiles");
    bench(conv1_img_kernel<1, 4, 8>, 8, "8 waves x 1 tile (2 passes)");
    bench(conv1_img_kernel<1, 4, 16>, 16, "16 waves x 1 tile");
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
    printf("   max |err| %.3g (max |ref| %.3g), %zu of %zu outside 1e-5: %s\n", worst, big, bad, ny, bad ? "FAIL" : "ok");
    return 0;
}
