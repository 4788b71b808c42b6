// Prototype (round 3): "image-stationary" convolution on the bf16 matrix pipe with exact three-way bf16 splits.
//   * one 512-thread workgroup keeps ONE input image in LDS as three pre-split bf16 planes (split once per element,
//     not once per tap), NHWC, 16-byte chunks XOR-swizzled;
//   * weights arrive pre-split in MFMA-fragment order (wprep kernel) and stream through a two-stage LDS ring, one
//     32-channel k-block (12 KB at 64 output columns) per barrier;
//   * wave (rt, nh) owns row tile rt (32 output pixels) x column half nh (32 channels): 9 MFMAs per 16 k, no vector
//     work in the loop except the tap's address.
// Checks against a float64 contraction and times it.  usage: img_conv_proto [images] [variant]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <type_traits>

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

struct ImgProblem {
    int rows, out_w;            // output pixels per image, output width
    int mul, add_y, add_x;      // tap origin of output pixel (oy, ox): (oy * mul + add_y, ox * mul + add_x)
    int taps_y, taps_x, step;   // tap (ty, tx) reads input pixel (y0 + step * ty, x0 + step * tx)
    int omul, oadd_y, oadd_x;   // written to output pixel (oy * omul + oadd_y, ox * omul + oadd_x)
    int w_blk;                  // first k-block of this problem in the fragment-ordered weights
};
struct ImgConvArgs {
    const float* x; float* y; const float* bias; const float* mask;
    const u32x4* wf;
    int n_img, H, W, OH, OW, N, relu, n_prob;
    int plane_bytes;            // bytes of one LDS plane (image + zero pixel, whole 16-row groups)
    int n_blocks;               // k-blocks of one image's schedule (the weights hold them in consumption order)
    unsigned long long* trace;
    ImgProblem p[4];
};

// fragment-ordered, pre-split weights: block kb (32 reduction indices) = [nh][step][plane][lane] 16-byte fragments;
// lane (l31, half) of (nh, step) holds output column n = 32 nh + l31, reduction indices kb*32 + step*16 + half*8 + 0..7
// forward layout: w[n][K] (K = taps * C, index (ty * kw + tx) * C + c)
__global__ void wprep_fwd_kernel(const float* w, u32x4* wf, int N, int K) {
    const int NH = N / 32;
    const int total = (K / 32) * NH * 2 * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63, s = (i >> 6) & 1, t = i >> 7, nh = t % NH, kb = t / NH;
        const int n = nh * 32 + (lane & 31), k0 = kb * 32 + s * 16 + (lane >> 5) * 8;
        const float* src = w + (size_t)n * K + k0;
        unsigned h[4], m[4], l[4];
        for (int q = 0; q < 4; ++q) split_pair(src[2 * q], src[2 * q + 1], h[q], m[q], l[q]);
        u32x4* dst = wf + ((size_t)(kb * NH + nh) * 2 + s) * 3 * 64 + lane;
        dst[0] = u32x4{h[0], h[1], h[2], h[3]}; dst[64] = u32x4{m[0], m[1], m[2], m[3]}; dst[128] = u32x4{l[0], l[1], l[2], l[3]};
    }
}

// swizzled byte offset (inside a plane) of 16-byte chunk L (= pixel * C/8 + chunk of the pixel): 128-byte rows of 8
// chunks, chunk column XORed with the row, adjacent rows swapped in every other group of 8 rows -- a ds_read_b128 of 16
// lanes at consecutive (or stride-2) pixels then touches 16 different 16-byte bank groups
__device__ __forceinline__ unsigned swz(unsigned L) {
    const unsigned row = L >> 3, col = L & 7;
    return ((row ^ ((row >> 3) & 1)) << 7) + ((col ^ (row & 7)) << 4);
}
// V5 layout: blocks of 16 pixels, chunk-major inside a block -- byte offset of 16-byte chunk c of pixel p (CH chunks per
// pixel): 16 lanes at consecutive pixels (any alignment) read 16 different bank groups; lanes whose tap is outside the
// image read slot (p & 15) of a ZERO block (p = the pixel index they would have had): the same bank group as inside
template <int CH> __device__ __forceinline__ unsigned blk16(unsigned p, unsigned c) {
    return (p >> 4) * (CH * 256) + c * 256 + (p & 15) * 16;
}

// V3: flattened block schedule (the fragment-ordered weights hold the blocks in consumption order), a THREE-stage weight
// ring filled two blocks ahead by LDS-DMA, the fragments of block j + 1 (image AND weights) read into registers during
// block j -- the barrier only guards the reuse of a ring stage.  The two waves that share a SIMD (w and w + 4: the column
// halves of one row tile) run the block's two parts in OPPOSITE order: role 0 issues its 18 MFMAs first and its loads after,
// role 1 the loads first -- both hit every barrier together, so with the same order their load stretches would coincide
// and leave the matrix pipe idle once per block.
struct Frag { u32x4 a[2][3], w[2][3]; };

template <int C, int NH, int WPC, bool DMA, int KO = 0>
__global__ __launch_bounds__(512, 2 * WPC) void img_conv_kernel(const ImgConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int CH = C / 8;                       // 16-byte chunks per pixel and plane
    constexpr int CB = C / 32;                      // k-blocks per tap
    constexpr int RTP = 8 / NH;                     // row tiles per pass
    constexpr int WBLK = NH * 2 * 3 * 1024;         // bytes of one weight k-block
    constexpr int PIECES = WBLK / 1024;             // 1 KB DMA pieces per block, dealt to the four role-1 waves
    static_assert(NH == 2 && PIECES % 4 == 0, "two column halves");
    constexpr int MAXLD = 8;                        // float4 loads per thread that cover an image
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int rt = wave % RTP, nh = wave / RTP, role = nh;
    const int H = a.H, W = a.W, npix = H * W, plane_bytes = a.plane_bytes, T = a.n_blocks;
    char* const sA = lds;
    const unsigned sW = 3 * plane_bytes;            // LDS byte offset of the ring
    unsigned long long t0 = 0, t1 = 0, t2 = 0, tk0 = 0, tk1 = 0, tk2 = 0;
    if (a.trace) t0 = __builtin_readcyclecounter();
    const int zero_base = (npix + 15) & ~15;        // the zero block: 16 pixel slots behind the image
    if (tid < 3 * CH * 16) {
        const int pl = tid / (CH * 16), c = (tid / 16) % CH;
        *reinterpret_cast<u32x4*>(sA + pl * plane_bytes + blk16<CH>(zero_base + (tid & 15), c)) = u32x4{0, 0, 0, 0};
    }
    // ---- weight ring: block G of this workgroup's stream is block G % T of the schedule, stage G % 3
    auto w_issue = [&](int blk, unsigned stage_off) {       // role-1 waves: PIECES / 4 pieces each
        const u32x4* src = a.wf + (size_t)blk * (WBLK / 16) + lane;
#pragma unroll
        for (int i = 0; i < PIECES / 4; ++i) {
            const int pc = (wave & 3) + 4 * i;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + pc * 64),
                                             (__attribute__((address_space(3))) void*)(lds + stage_off + pc * 1024), 16, 0, 0);
        }
    };
    // ---- image loads: all issued first, split + stored afterwards
    float4 ireg[MAXLD];
    const int n4 = npix * (C / 4);
    auto img_issue = [&](int img) {
        const float4* src = reinterpret_cast<const float4*>(a.x) + (size_t)img * n4;
#pragma unroll
        for (int i = 0; i < MAXLD; ++i)
            if (tid + 512 * i < n4) ireg[i] = src[tid + 512 * i];
    };
    auto img_store = [&]() {
#pragma unroll
        for (int i = 0; i < MAXLD; ++i) {
            const int idx = tid + 512 * i;
            if (idx < n4) {
                const float4 v = ireg[i];
                unsigned h0, m0, l0, h1, m1, l1;
                split_pair(v.x, v.y, h0, m0, l0);
                split_pair(v.z, v.w, h1, m1, l1);
                const unsigned off = blk16<CH>((unsigned)idx / (C / 4), ((unsigned)idx % (C / 4)) >> 1) + ((idx & 1) << 3);
                *reinterpret_cast<u32x2*>(sA + off) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(sA + plane_bytes + off) = u32x2{m0, m1};
                *reinterpret_cast<u32x2*>(sA + 2 * plane_bytes + off) = u32x2{l0, l1};
            }
        }
    };
    // ring state (scalar): byte offsets of the stages holding blocks G, G + 1, G + 2; schedule index of block G + 2
    unsigned st0 = sW, st1 = sW + WBLK, st2 = sW + 2 * WBLK;
    int b2 = 2 % T;
    img_issue(blockIdx.x);
    if (role == 1) { w_issue(0, st0); w_issue(1 % T, st1); }
    for (int img = blockIdx.x; img < a.n_img; img += gridDim.x) {
        __syncthreads();                            // the previous image's fragment reads are done
        img_store();
        if (img + (int)gridDim.x < a.n_img) img_issue(img + gridDim.x);
        __syncthreads();
        if (a.trace && img == (int)blockIdx.x) t1 = __builtin_readcyclecounter();
        for (int pr = 0; pr < a.n_prob; ++pr) {
            const ImgProblem P = a.p[pr];
            const int taps_x = P.taps_x, ntaps = P.taps_y * P.taps_x, nkb = ntaps * CB, tap_step = P.step, row_step = P.step * W;
            for (int m0 = 0; m0 < P.rows; m0 += RTP * 32) {
                // ---- this lane's output pixel, its tap origin and which taps fall outside the image
                const int m = m0 + rt * 32 + l31;
                const bool row_ok = m < P.rows;
                const int oy = m / P.out_w, ox = m - oy * P.out_w;
                const int y0 = oy * P.mul + P.add_y, x0 = ox * P.mul + P.add_x;
                unsigned bad = 0;
                for (int ty = 0; ty < P.taps_y; ++ty)
                    for (int tx = 0; tx < taps_x; ++tx) {
                        const int yy = y0 + P.step * ty, xx = x0 + P.step * tx;
                        const bool ok = row_ok && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
                        if (!ok) bad |= 1u << (ty * taps_x + tx);
                    }
                const int pbase = y0 * W + x0;
                f32x16 acc;
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[v] = 0.f;
                // scalar tap state of the block whose fragments are read NEXT: tap index, column, pixel offset, channel block
                int tap = 0, tx = 0, toff = 0, rowoff = 0, cb = 0;
                Frag F[2];
                auto read_frags = [&](Frag& f, unsigned stage_off) {
                    const int lin = pbase + toff;
                    const int pix = ((bad >> tap) & 1) ? zero_base + (lin & 15) : lin;
                    const unsigned o0 = blk16<CH>((unsigned)pix, (unsigned)(cb * 4 + half));
                    const char* wS = lds + stage_off + (nh * 2) * 3072 + lane * 16;
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) {
                            f.a[s][pl] = *reinterpret_cast<const u32x4*>(sA + pl * plane_bytes + o0 + s * 512);
                            f.w[s][pl] = *reinterpret_cast<const u32x4*>(wS + s * 3072 + pl * 1024);
                        }
                    if (++cb == CB) {
                        cb = 0; ++tap; ++tx; toff += tap_step;
                        if (tx == taps_x) { tx = 0; rowoff += row_step; toff = rowoff; }
                        if (tap == ntaps) { tap = 0; rowoff = 0; toff = 0; }
                    }
                };
                auto mfmas = [&](const Frag& f) {
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int sum = 4; sum >= 0; --sum)
#pragma unroll
                            for (int pa = 0; pa < 3; ++pa) {
                                const int pb = sum - pa;
                                if (pb < 0 || pb > 2) continue;
                                acc = mfma_bf16(f.w[s][pb], f.a[s][pa], acc);
                            }
                };
                read_frags(F[0], st0);              // block G landed before the last barrier
                if (a.trace && img == (int)blockIdx.x) tk0 = __builtin_readcyclecounter();
                // One block = one straight-line region: [barrier] the next block's weight DMA (role 1), the fragment reads of
                // block G + 1 and the 18 MFMAs of block G, dealt out by the group barriers below: the reads go out after a few
                // MFMAs and land under the rest; the two roles place them at different depths of the MFMA stream.
                // (reads and DMA are unconditional: past the end of a pass they fetch this pass's first tap again / the
                //  schedule's next block, both valid)
                auto block = [&](auto role_c, Frag& cur, Frag& nxt) {
                    constexpr int ROLE = decltype(role_c)::value;
                    if (!(KO & 1)) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();   // block G + 1 is in its stage; nobody reads stage st2 any more
                    }
                    if (ROLE == 1 && !(KO & 4)) w_issue(b2, st2);
                    if (!(KO & 2)) read_frags(nxt, st1);
                    mfmas(cur);
                    const unsigned t = st0; st0 = st1; st1 = st2; st2 = t;
                    b2 = b2 + 1 == T ? 0 : b2 + 1;
                    if (!(KO & 8)) {
                    constexpr int LEAD = ROLE == 0 ? 2 : 7;         // MFMAs ahead of the address math
                    __builtin_amdgcn_sched_group_barrier(0x008, LEAD, 0);
                    if (ROLE == 1) __builtin_amdgcn_sched_group_barrier(0x020, PIECES / 4, 0);
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 18 - LEAD - 7, 0);
                    }
                };
                auto k_loop = [&](auto role_c) {
                    for (int kb = 0; kb < nkb; kb += 2) { block(role_c, F[0], F[1]); block(role_c, F[1], F[0]); }
                };
                if (role == 0) k_loop(std::integral_constant<int, 0>{}); else k_loop(std::integral_constant<int, 1>{});
                if (a.trace && img == (int)blockIdx.x) tk1 = __builtin_readcyclecounter();
                // ---- epilogue: lane = output pixel l31, channels 32 nh + 8 q + 4 half + 0..3
                if (row_ok) {
                    const size_t orow = ((size_t)(img * a.OH + oy * P.omul + P.oadd_y) * a.OW + ox * P.omul + P.oadd_x) * a.N;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = nh * 32 + 8 * q + 4 * half;
                        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (a.bias) b = *reinterpret_cast<const float4*>(a.bias + n);
                        float4 v = make_float4(acc[4 * q] + b.x, acc[4 * q + 1] + b.y, acc[4 * q + 2] + b.z, acc[4 * q + 3] + b.w);
                        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        if (a.mask) {
                            const float4 mk = *reinterpret_cast<const float4*>(a.mask + orow + n);
                            if (!(mk.x > 0.f)) v.x = 0.f;
                            if (!(mk.y > 0.f)) v.y = 0.f;
                            if (!(mk.z > 0.f)) v.z = 0.f;
                            if (!(mk.w > 0.f)) v.w = 0.f;
                        }
                        *reinterpret_cast<float4*>(a.y + orow + n) = v;
                    }
                }
            }
        }
        if (a.trace && img == (int)blockIdx.x) t2 = __builtin_readcyclecounter();
    }
    if (a.trace && (tid & 63) == 0) {
        unsigned long long* t = a.trace + (blockIdx.x * 8 + wave) * 8;
        t[0] = t0; t[1] = t1; t[2] = t2; t[3] = __builtin_readcyclecounter(); t[4] = tk0; t[5] = tk1;
    }
}

// float64 reference: forward convolution NHWC, weights (N, kh, kw, C)
__global__ void ref_conv_kernel(const float* x, const float* w, const float* bias, double* y, int n_img, int H, int W, int C,
                                int N, int kh, int kw, int stride, int pad, int OH, int OW) {
    const size_t total = (size_t)n_img * OH * OW * N;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = i % N; size_t t = i / N;
        const int ox = t % OW; t /= OW;
        const int oy = t % OH; const int b = t / OH;
        double s = 0;
        for (int ty = 0; ty < kh; ++ty)
            for (int tx = 0; tx < kw; ++tx) {
                const int yy = oy * stride + ty - pad, xx = ox * stride + tx - pad;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                const float* xp = x + ((size_t)(b * H + yy) * W + xx) * C;
                const float* wp = w + ((size_t)n * kh * kw + ty * kw + tx) * C;
                for (int c = 0; c < C; ++c) s += (double)xp[c] * (double)wp[c];
            }
        s += bias[n];
        y[i] = s > 0 ? s : 0;
    }
}

template <int C, int NH, int WPC, bool DMA, int KO = 0>
float run(const char* what, ImgConvArgs a, int grid, size_t lds_bytes, int reps) {
    auto k = img_conv_kernel<C, NH, WPC, DMA, KO>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds_bytes, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds_bytes, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s grid %4d lds %6zu: %.2f us per launch\n", what, grid, lds_bytes, ms / reps * 1e3);
    return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
    const int n_img = argc > 1 ? atoi(argv[1]) : 512;
    struct Layer { const char* name; int H, W, C, N, k, stride, pad; };
    const Layer layers[2] = {{"conv2 fwd (25x19x32 -> 12x9x64, 4x4 s2 p1)", 25, 19, 32, 64, 4, 2, 1},
                             {"conv3 fwd (12x9x64 -> 12x9x64, 3x3 s1 p1)", 12, 9, 64, 64, 3, 1, 1}};
    for (int li = 0; li < 2; ++li) {
        const Layer L = layers[li];
        const int OH = (L.H + 2 * L.pad - L.k) / L.stride + 1, OW = (L.W + 2 * L.pad - L.k) / L.stride + 1;
        const int K = L.k * L.k * L.C;
        const size_t nx = (size_t)n_img * L.H * L.W * L.C, ny = (size_t)n_img * OH * OW * L.N, nw = (size_t)L.N * K;
        std::vector<float> hx(nx), hw(nw), hb(L.N);
        srand(1 + li);
        for (auto& v : hx) { float r = (rand() / (float)RAND_MAX) * 2.f - 0.8f; v = r > 0 ? r : 0; }
        for (auto& v : hw) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.2f;
        for (auto& v : hb) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.1f;
        float *dx, *dw, *db, *dy; double* dref; u32x4* dwf;
        CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&db, L.N * 4)); CK(hipMalloc(&dy, ny * 4));
        CK(hipMalloc(&dref, ny * 8)); CK(hipMalloc(&dwf, nw * 6));
        CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb.data(), L.N * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(wprep_fwd_kernel, dim3(64), dim3(256), 0, 0, dw, dwf, L.N, K);
        hipLaunchKernelGGL(ref_conv_kernel, dim3(2048), dim3(256), 0, 0, dx, dw, db, dref, n_img, L.H, L.W, L.C, L.N, L.k, L.k,
                           L.stride, L.pad, OH, OW);
        CK(hipDeviceSynchronize());
        ImgConvArgs a = {};
        a.x = dx; a.y = dy; a.bias = db; a.mask = nullptr; a.wf = dwf;
        a.n_img = n_img; a.H = L.H; a.W = L.W; a.OH = OH; a.OW = OW; a.N = L.N; a.relu = 1; a.n_prob = 1;
        a.plane_bytes = ((L.H * L.W + 15) / 16 + 1) * 16 * (L.C / 8) * 16;
        a.p[0] = ImgProblem{OH * OW, OW, L.stride, -L.pad, -L.pad, L.k, L.k, 1, 1, 0, 0, 0};
        a.n_blocks = K / 32;
        const size_t lds_bytes = 3 * (size_t)a.plane_bytes + 3 * (L.N / 32) * 2 * 3 * 1024;
        printf("== %s, %d images: K %d, plane %d B, %.1f GF fp32 (x9 bf16)\n", L.name, n_img, K, a.plane_bytes,
               2.0 * n_img * OH * OW * L.N * K * 1e-9);
        const int cus = 256;
        for (int variant = 0; variant < 4; ++variant) {
            CK(hipMemset(dy, 0xff, ny * 4));
            float us = 0;
            const int wpc = (variant & 1) ? 2 : 1;
            if (wpc == 2 && 2 * lds_bytes > 160 * 1024) continue;
            const int grid = n_img < cus * wpc ? n_img : cus * wpc;
            if (L.C == 32) {
                if (variant == 0) us = run<32, 2, 1, false>("plain weight copies, 1 workgroup per CU", a, grid, lds_bytes, 50);
                if (variant == 1) us = run<32, 2, 2, false>("plain weight copies, 2 workgroups per CU", a, grid, lds_bytes, 50);
                if (variant == 2) us = run<32, 2, 1, true>("LDS-DMA weights, 1 workgroup per CU", a, grid, lds_bytes, 50);
                if (variant == 3) us = run<32, 2, 2, true>("LDS-DMA weights, 2 workgroups per CU", a, grid, lds_bytes, 50);
            } else {
                if (variant == 0) us = run<64, 2, 1, false>("plain weight copies, 1 workgroup per CU", a, grid, lds_bytes, 50);
                if (variant == 1) us = run<64, 2, 2, false>("plain weight copies, 2 workgroups per CU", a, grid, lds_bytes, 50);
                if (variant == 2) us = run<64, 2, 1, true>("LDS-DMA weights, 1 workgroup per CU", a, grid, lds_bytes, 50);
                if (variant == 3) us = run<64, 2, 2, true>("LDS-DMA weights, 2 workgroups per CU", a, grid, lds_bytes, 50);
            }
            {   // phase cycles of every workgroup's first image (one extra launch with timestamps)
                unsigned long long* dtr; CK(hipMalloc(&dtr, (size_t)grid * 512));
                ImgConvArgs at = a; at.trace = dtr;
                if (L.C == 32) { if (variant < 2) run<32, 2, 1, false>("  (traced)", at, grid, lds_bytes, 1); else run<32, 2, 1, true>("  (traced)", at, grid, lds_bytes, 1); }
                else { if (variant < 2) run<64, 2, 1, false>("  (traced)", at, grid, lds_bytes, 1); else run<64, 2, 1, true>("  (traced)", at, grid, lds_bytes, 1); }
                std::vector<unsigned long long> tr((size_t)grid * 64);
                CK(hipMemcpy(tr.data(), dtr, (size_t)grid * 512, hipMemcpyDeviceToHost));
                double p0 = 0, p1 = 0, p2 = 0, q0 = 0, q1 = 0, q2 = 0;
                for (int i = 0; i < grid * 8; ++i) {
                    const unsigned long long* t = &tr[8 * i];
                    p0 += t[1] - t[0]; p1 += t[2] - t[1]; p2 += t[3] - t[0]; q0 += t[4] - t[1]; q1 += t[5] - t[4]; q2 += t[2] - t[5];
                }
                const double n = grid * 8.0;
                printf("   cycles per wave: image load + split %.0f, first image: pass prologue %.0f, k-loop %.0f, epilogue %.0f (sum %.0f); whole %.0f\n",
                       p0 / n, q0 / n, q1 / n, q2 / n, p1 / n, p2 / n);
                CK(hipFree(dtr));
            }
            if (variant == 2 && L.C == 32) {
                run<32, 2, 1, true, 1>("   KO: no barrier", a, grid, lds_bytes, 50);
                run<32, 2, 1, true, 2>("   KO: no fragment reads in the loop", a, grid, lds_bytes, 50);
                run<32, 2, 1, true, 4>("   KO: no weight DMA", a, grid, lds_bytes, 50);
                run<32, 2, 1, true, 7>("   KO: none of the three", a, grid, lds_bytes, 50);
                run<32, 2, 1, true, 3>("   KO: no barrier, no fragment reads", a, grid, lds_bytes, 50);
                run<32, 2, 1, true, 8>("   KO: no group barriers (compiler's order)", a, grid, lds_bytes, 50);
                run<32, 2, 1, true, 0>("   (restored)", a, grid, lds_bytes, 50);
            }
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
            printf("   max |err| %.3g (max |ref| %.3g), %zu of %zu outside 1e-5: %s; %.0f TF/s fp32-equivalent\n", worst, big, bad, ny,
                   bad ? "FAIL" : "ok", 2.0 * n_img * OH * OW * L.N * K / us * 1e-6);
        }
        CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dy)); CK(hipFree(dref)); CK(hipFree(dwf));
    }
    return 0;
}
