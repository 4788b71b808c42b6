// "Image-stationary" kernels (round 3): a persistent workgroup keeps what a whole IMAGE's outputs need in LDS instead of
// gathering it tap by tap from global memory.
//
// conv1_img_kernel: the first convolution straight from the sampler's u8 observations (arl_conv2d_u8_fwd; the
// reference's network input is obs * (1 / 255): accel_rl/policies/pg/atari_cnn_policy.py:88-91, first layer
// pg_cnn.py:47-52) for 32 filters of 8 x 8 pixels:
//   * the layer's weights are split ONCE per workgroup -- exactly, into three bf16 planes (mfma_common.h, SPLIT) --
//     and stay in LDS in MFMA-fragment order for the workgroup's whole life (48 KB at 4 planes);
//   * an image (33 KB of u8) arrives with coalesced 16-byte loads into one of two LDS buffers while the previous image
//     is being computed; one barrier per image;
//   * an MFMA A fragment -- eight consecutive pixels of one filter row -- is two dwords of LDS, converted to bf16
//     (0 .. 255 is exact in one piece) between the read and the MFMA: three products per multiply, as in igemm_body's
//     U8 path, in the same order (weight planes l, m, h per 16-k step; steps in (plane, filter row) order), so the
//     results are bit-identical to it;
//   * 16 waves per workgroup, a wave owns one 32-pixel row tile of an image at a time.
// The kernel it replaces gathers one dword per lane and instruction from global memory (64 cache lines per load
// instruction: bound by the texture addresser, 24.5 us at the PPO minibatch); this one runs 19 us there and 10.9 us
// instead of 15.6 us at the rollout's 256 rows (tools/proto/conv1_img_proto.hip, profiles/r03/conv1_img_proto.txt).
#include "mfma_conv_impl.h"
#include "img_conv_dev.h"

namespace arlc {

__global__ __launch_bounds__(C1_NT) void conv1_img_kernel(const Conv1ImgArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int MAXLD = (C1_MAX_IMG / 16 + C1_NT - 1) / C1_NT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int C = a.C, H = a.H, W = a.W, npix = C * H * W, K = C * 64, nsteps = K / 16;
    char* const sW = lds;                                   // [nsteps][3 planes][64 lanes] 16-byte fragments
    char* const sI = lds + nsteps * 3072;                   // two u8 images [C][H][W]
    const int n16 = npix / 16;
    u32x4 ireg[MAXLD];
    auto img_issue = [&](int img) {
        int row = a.idx ? a.idx[img] : img;
        row = (unsigned)row < (unsigned)a.obs_rows ? row : 0;
        const u32x4* src = reinterpret_cast<const u32x4*>(a.obs + (size_t)row * npix);
#pragma unroll
        for (int i = 0; i < MAXLD; ++i)
            if (tid + C1_NT * i < n16) ireg[i] = src[tid + C1_NT * i];
    };
    auto img_store = [&](int buf, int img) {
        char* d = sI + buf * npix;
#pragma unroll
        for (int i = 0; i < MAXLD; ++i)
            if (tid + C1_NT * i < n16) *reinterpret_cast<u32x4*>(d + (tid + C1_NT * i) * 16) = ireg[i];
        if (a.copy_out) {
            u32x4* o = reinterpret_cast<u32x4*>(a.copy_out + (long long)img * a.copy_stride);
#pragma unroll
            for (int i = 0; i < MAXLD; ++i)
                if (tid + C1_NT * i < n16) o[tid + C1_NT * i] = ireg[i];
        }
    };
    if (a.zero_word && blockIdx.x == 0 && tid == 0) *a.zero_word = 0;
    img_issue(blockIdx.x);
    for (int f = tid; f < nsteps * 64; f += C1_NT) conv1_w_store(sW, f, conv1_w_load(a.w, K, f, a.NF));
    const int rows = a.OH * a.OW, tiles = (rows + 31) / 32;
    float4 bq[4];
    conv1_bias_quads(a.bias, half, bq, a.NF);
    img_store(0, blockIdx.x);
    if (blockIdx.x + gridDim.x < (unsigned)a.n_img) img_issue(blockIdx.x + gridDim.x);
    __syncthreads();
    int buf = 0;
    for (int img = blockIdx.x; img < a.n_img; img += gridDim.x, buf ^= 1) {
        const char* im = sI + buf * npix;
        for (int tp = wave; tp < tiles; tp += C1_NW) conv1_tile(a, im, sW, img, tp, lane, bq);
        // the next image's bytes (in flight while this one was computed) -> the other buffer, which every wave left at
        // the last barrier; the image after that starts its way from memory
        if (img + (int)gridDim.x < a.n_img) {
            img_store(buf ^ 1, img + (int)gridDim.x);
            if (img + 2 * (int)gridDim.x < a.n_img) img_issue(img + 2 * gridDim.x);
        }
        __syncthreads();
    }
}

bool g_no_img_kernels = false;          // arl_dev_conv_variant(1): the tap-gathering kernels everywhere (A/B, parity tests)

// arl_conv2d_u8_fwd's geometries that take the image-stationary kernel; < 0: not one of them (nothing launched)
int launch_conv1_img(const unsigned char* obs, int64_t obs_rows, const int32_t* idx, float scale, const float* w, const float* bias, float* y,
                     int64_t batch, int C, int H, int W, int K, int kh, int kw, int stride, int Ho, int Wo, int relu,
                     hipStream_t s, unsigned char* copy_out, long long copy_stride, int* zero_word) {
    const int npix = C * H * W;
    const size_t lds = (size_t)(C * 64 / 16) * 3072 + 2 * (size_t)npix;
    if (g_no_img_kernels || !t_ctx.split || t_ctx.split == 1 || (K != 32 && K != 16) || kh != 8 || kw != 8 || npix % 16 != 0 || npix > C1_MAX_IMG ||
        lds > 160 * 1024 || ((uintptr_t)obs & 15) || (W & 3) || (stride & 3) || batch > 0x7fffffff)
        return -1;
    Conv1ImgArgs a = {obs, idx, w, bias, y, scale, (int)batch, C, H, W, Ho, Wo, stride, relu, K,
                      (int)(obs_rows < 0x7fffffff ? obs_rows : 0x7fffffff), copy_out, copy_stride, zero_word};
    {   // per launch: the attribute belongs to the CURRENT device's copy of the kernel (no process-wide flag)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_img_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { arl::set_error("hipFuncSetAttribute(conv1_img_kernel): %s", hipGetErrorString(e)); return (int)e; }
    }
    const int cus = 256;
    hipLaunchKernelGGL(conv1_img_kernel, dim3((unsigned)(batch < cus ? batch : cus)), dim3(C1_NT), lds, s, a);
    return arl::check_launch("conv1_img_kernel");
}

// arl_rollout_begin_conv1 (serve_step.hip): the same launch under the call's route, with the copy-out
int launch_conv1_img_begin(const arl_conv_geom* geom, const unsigned char* obs, int64_t n_img, float scale, const float* w,
                           const float* bias, float* y, int C, int relu, hipStream_t s, unsigned char* copy_out,
                           long long copy_stride, int* zero_word) {
    ARL_ROUTE_SCOPE(geom, nullptr);
    const int st = geom->stride;
    return launch_conv1_img(obs, n_img, nullptr, scale, w, bias, y, n_img, C, geom->in_h, geom->in_w, geom->out_c, geom->kh,
                            geom->kw, st, (geom->in_h - geom->kh) / st + 1, (geom->in_w - geom->kw) / st + 1, relu, s, copy_out,
                            copy_stride, zero_word);
}

}  // namespace arlc

extern "C" void arl_dev_conv_variant(int32_t v) { arlc::g_no_img_kernels = (v & 1) != 0; }
