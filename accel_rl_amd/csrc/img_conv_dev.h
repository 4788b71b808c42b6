// Device-side pieces of the image-stationary first convolution (img_conv.hip: conv1_img_kernel; serve_step.hip: the same
// tiles on the observation the env step has just left in LDS).  Both callers issue the same piece products in the same
// order: bit-identical outputs.
#pragma once
#include "mfma_common.h"

namespace arlc {

struct Conv1ImgArgs {
    const unsigned char* obs;   // u8 [rows][C][H][W]
    const int* idx;             // row of image b, or null
    const float* w;             // f32 [32][C][8][8]
    const float* bias;          // f32 [32] or null
    float* y;                   // f32 [B][OH][OW][32]
    float scale;
    int n_img, C, H, W, OH, OW, stride, relu;
    int NF;                     // filters: 32, or 16 (the upper half of the 32-row MFMA tile then multiplies zeros: at three
                                // piece products per multiply on the bf16 pipe still 2.7 x the fp32 chain's rate)
    int obs_rows;               // rows of obs: an index outside [0, obs_rows) reads row 0 instead of faulting
    // arl_rollout_begin_conv1: image b is also copied, as it passes through the registers, to copy_out + b * copy_stride
    // (the rollout buffer's row (env b, step 0)) and *zero_word = 0 (the completed-trajectory counter); null: neither
    unsigned char* copy_out;
    long long copy_stride;
    int* zero_word;
};

__device__ __forceinline__ u32x2 bytes_to_bf16x4(unsigned v) {     // four packed bytes -> four bf16, exact
    const float4 f = bytes_to_f4(v);
    return u32x2{hi_pair(f.x, f.y), hi_pair(f.z, f.w)};
}

// Weight fragment f = (step s = f >> 6, lane fl = f & 63) of the LDS image [nsteps][3 planes][64 lanes] x 16 bytes:
// filter fl & 31, reduction indices 16 s + 8 (fl >> 5) + 0..7 (index (c * 8 + ty) * 8 + tx), split exactly into three
// bf16 planes.  Load and split + store are separate so that a caller can put other work under the load's latency.
struct Conv1WFrag { float4 v0, v1; };
__device__ __forceinline__ Conv1WFrag conv1_w_load(const float* __restrict__ w, const int K, const int f, const int NF = 32) {
    const int s = f >> 6, fl = f & 63;
    if ((fl & 31) >= NF) return Conv1WFrag{make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    const float* src = w + (size_t)(fl & 31) * K + s * 16 + (fl >> 5) * 8;
    return Conv1WFrag{*reinterpret_cast<const float4*>(src), *reinterpret_cast<const float4*>(src + 4)};
}
__device__ __forceinline__ void conv1_w_store(char* sW, const int f, const Conv1WFrag& v) {
    const int s = f >> 6, fl = f & 63;
    unsigned h[4], m[4], l[4];
    split_pair(v.v0.x, v.v0.y, h[0], m[0], l[0]);
    split_pair(v.v0.z, v.v0.w, h[1], m[1], l[1]);
    split_pair(v.v1.x, v.v1.y, h[2], m[2], l[2]);
    split_pair(v.v1.z, v.v1.w, h[3], m[3], l[3]);
    char* d = sW + s * 3072 + fl * 16;
    *reinterpret_cast<u32x4*>(d) = u32x4{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<u32x4*>(d + 1024) = u32x4{m[0], m[1], m[2], m[3]};
    *reinterpret_cast<u32x4*>(d + 2048) = u32x4{l[0], l[1], l[2], l[3]};
}

// the lane's bias quads (channels 8 q + 4 half + 0..3)
__device__ __forceinline__ void conv1_bias_quads(const float* __restrict__ bias, const int half, float4 (&bq)[4],
                                                 const int NF = 32) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias && 8 * q + 4 * half < NF) bq[q] = *reinterpret_cast<const float4*>(bias + 8 * q + 4 * half);
    }
}

// One 32-pixel row tile tp of image `img` whose u8 planes [C][H][W] sit at `im` in LDS: 16-k steps in (plane, filter
// row) order, weight planes l, m, h per step (split_products<.., 1, 3, true>), scale + bias + rectifier, 16-byte stores.
__device__ __forceinline__ void conv1_tile(const Conv1ImgArgs& a, const char* im, const char* sW, const int img,
                                           const int tp, const int lane, const float4 (&bq)[4]) {
    constexpr int KH = 8;
    const int N = a.NF;
    const int l31 = lane & 31, half = lane >> 5;
    const int H = a.H, W = a.W, nsteps = a.C * 64 / 16;
    const int rows = a.OH * a.OW;
    const int m = tp * 32 + l31;
    const int mm = m < rows ? m : 0;
    const int oy = mm / a.OW, ox = mm - oy * a.OW;
    const unsigned off = (unsigned)((oy * a.stride + half) * W + ox * a.stride);     // filter row `half` of a step
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    const char* wl = sW + lane * 16;
#pragma unroll 2
    for (int s = 0; s < nsteps; ++s) {              // step s = plane s / 4, filter rows 2 (s % 4) + half
        const unsigned po = (unsigned)(((s / (KH / 2)) * H + 2 * (s % (KH / 2))) * W);
        u32x4 fb[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) fb[pl] = *reinterpret_cast<const u32x4*>(wl + s * 3072 + pl * 1024);
        const unsigned* q = reinterpret_cast<const unsigned*>(im + po + off);
        const u32x2 lo = bytes_to_bf16x4(q[0]), hi = bytes_to_bf16x4(q[1]);
        const u32x4 fa = u32x4{lo.x, lo.y, hi.x, hi.y};
#pragma unroll
        for (int pl = 2; pl >= 0; --pl) acc = mfma_bf16(fb[pl], fa, acc);   // (split_products<.., 1, 3, true>: l, m, h)
    }
    if (m < rows) {                                 // lane = pixel m, channels 8 q + 4 half + 0..3
        float* dst = a.y + ((size_t)img * rows + m) * N + 4 * half;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (8 * q + 4 * half >= N) continue;            // (16 filters: the tile's upper rows are padding)
            float4 v = make_float4(acc[4 * q] * a.scale + bq[q].x, acc[4 * q + 1] * a.scale + bq[q].y,
                                   acc[4 * q + 2] * a.scale + bq[q].z, acc[4 * q + 3] * a.scale + bq[q].w);
            if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(dst + 8 * q) = v;
        }
    }
}

constexpr int C1_NW = 16, C1_NT = C1_NW * 64, C1_MAX_IMG = 40960;

}  // namespace arlc
