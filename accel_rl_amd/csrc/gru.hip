// GRU and plain tanh-RNN cells for the recurrent policies: the elementwise part of
// GruLayer.step / RecurrentLayer.step, accel_rl/policies/layers.py:163-168, 80-82, and their
// backward.  As for the LSTM (csrc/lstm.hip) the matrix products x W_x + b (all steps at once) and
// h_prev W_h (per step) are dense fp32-MFMA calls; rows may be strided so that a time slice of a
// [trajectory][time] batch is addressed in place.
//
// GRU, gate order r, u, c in the 3H-wide arrays:
//   r = s(gx_r + gh_r); u = s(gx_u + gh_u); c = tanh(gx_c + r * gh_c); h = (1 - u) h_prev + u c
//   saved[B][4H] = r, u, c, gh_c
//   bwd: du = dh (c - h_prev); dc = dh u; dpc = dc (1 - c^2); dpu = du u (1 - u);
//        dpr = dpc gh_c r (1 - r);  dgx = [dpr, dpu, dpc];  dgh = [dpr, dpu, dpc r];
//        dh_prev (direct part) = dh (1 - u)        (+ dgh W_h^T, the caller's dense product)
// RNN: h = tanh(gx + gh); bwd: dpre = dh (1 - h^2)

#include "arl_common.h"

namespace {

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

struct GruArgs {
    const float* gx;        // [B][3H] strided
    const float* gh;        // [B][3H] contiguous
    const float* h_prev;    // [B][H] strided
    float* h_out;           // [B][H] strided
    float* saved;           // [B][4H] strided, or null
    int64_t batch;
    int hidden;
    int64_t gx_stride, hprev_stride, h_stride, saved_stride;
};

__global__ __launch_bounds__(256) void gru_fwd_kernel(const GruArgs a) {
    const int H = a.hidden;
    const int64_t total = a.batch * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / H;
        const int j = (int)(i - b * H);
        const float* gx = a.gx + b * a.gx_stride;
        const float* gh = a.gh + b * 3 * H;
        const float r = sigmoidf(gx[j] + gh[j]);
        const float u = sigmoidf(gx[H + j] + gh[H + j]);
        const float ghc = gh[2 * H + j];
        const float c = tanhf(gx[2 * H + j] + r * ghc);
        const float hp = a.h_prev[b * a.hprev_stride + j];
        a.h_out[b * a.h_stride + j] = (1.f - u) * hp + u * c;
        if (a.saved) {
            float* s = a.saved + b * a.saved_stride;
            s[j] = r; s[H + j] = u; s[2 * H + j] = c; s[3 * H + j] = ghc;
        }
    }
}

struct GruBwdArgs {
    const float* dh;        // [B][H] strided (layers above), or null
    const float* dh_rec;    // [B][H] contiguous (dgh W_h^T of step t+1), or null
    const float* dh_dir;    // [B][H] contiguous (direct part of step t+1), or null
    const float* saved;     // [B][4H] strided
    const float* h_prev;    // [B][H] strided
    float* dgx;             // [B][3H] strided
    float* dgh;             // [B][3H] strided
    float* dh_prev;         // [B][H] contiguous: direct part dh (1 - u)
    int64_t batch;
    int hidden;
    int64_t dh_stride, saved_stride, hprev_stride, dgx_stride, dgh_stride;
};

__global__ __launch_bounds__(256) void gru_bwd_kernel(const GruBwdArgs a) {
    const int H = a.hidden;
    const int64_t total = a.batch * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / H;
        const int j = (int)(i - b * H);
        const float* s = a.saved + b * a.saved_stride;
        const float r = s[j], u = s[H + j], c = s[2 * H + j], ghc = s[3 * H + j];
        float dh = 0.f;
        if (a.dh) dh += a.dh[b * a.dh_stride + j];
        if (a.dh_rec) dh += a.dh_rec[i];
        if (a.dh_dir) dh += a.dh_dir[i];
        const float hp = a.h_prev[b * a.hprev_stride + j];
        const float dpc = dh * u * (1.f - c * c);
        const float dpu = dh * (c - hp) * u * (1.f - u);
        const float dpr = dpc * ghc * r * (1.f - r);
        float* dgx = a.dgx + b * a.dgx_stride;
        float* dgh = a.dgh + b * a.dgh_stride;
        dgx[j] = dpr; dgx[H + j] = dpu; dgx[2 * H + j] = dpc;
        dgh[j] = dpr; dgh[H + j] = dpu; dgh[2 * H + j] = dpc * r;
        a.dh_prev[i] = dh * (1.f - u);
    }
}

struct RnnArgs {
    const float* gx;        // [B][H] strided
    const float* gh;        // [B][H] contiguous
    float* h_out;           // [B][H] strided
    int64_t batch;
    int hidden;
    int64_t gx_stride, h_stride;
};

__global__ __launch_bounds__(256) void rnn_fwd_kernel(const RnnArgs a) {
    const int H = a.hidden;
    const int64_t total = a.batch * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / H;
        const int j = (int)(i - b * H);
        a.h_out[b * a.h_stride + j] = tanhf(a.gx[b * a.gx_stride + j] + a.gh[i]);
    }
}

struct RnnBwdArgs {
    const float* dh;        // [B][H] strided, or null
    const float* dh_rec;    // [B][H] contiguous, or null
    const float* h_out;     // [B][H] strided
    float* dpre;            // [B][H] strided
    int64_t batch;
    int hidden;
    int64_t dh_stride, h_stride, dpre_stride;
};

__global__ __launch_bounds__(256) void rnn_bwd_kernel(const RnnBwdArgs a) {
    const int H = a.hidden;
    const int64_t total = a.batch * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / H;
        const int j = (int)(i - b * H);
        float dh = 0.f;
        if (a.dh) dh += a.dh[b * a.dh_stride + j];
        if (a.dh_rec) dh += a.dh_rec[i];
        const float h = a.h_out[b * a.h_stride + j];
        a.dpre[b * a.dpre_stride + j] = dh * (1.f - h * h);
    }
}

}  // namespace

extern "C" int arl_gru_cell_fwd(const float* gx, int64_t gx_stride, const float* gh, const float* h_prev,
                                int64_t hprev_stride, int64_t batch, int32_t hidden, float* h_out, int64_t h_stride,
                                float* saved_or_null, int64_t saved_stride, void* stream) {
    ARL_REQUIRE(gx && gh && h_prev && h_out, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(batch > 0 && hidden > 0, ARL_E_RANGE, "bad batch / hidden");
    GruArgs a = {gx, gh, h_prev, h_out, saved_or_null, batch, hidden, gx_stride, hprev_stride, h_stride, saved_stride};
    hipLaunchKernelGGL(gru_fwd_kernel, dim3(arl::stream_grid(batch * hidden, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return arl::check_launch("gru_fwd_kernel");
}

extern "C" int arl_gru_cell_bwd(const float* dh_or_null, int64_t dh_stride, const float* dh_rec_or_null,
                                const float* dh_dir_or_null, const float* saved, int64_t saved_stride,
                                const float* h_prev, int64_t hprev_stride, int64_t batch, int32_t hidden,
                                float* dgx, int64_t dgx_stride, float* dgh, int64_t dgh_stride, float* dh_prev,
                                void* stream) {
    ARL_REQUIRE(saved && h_prev && dgx && dgh && dh_prev, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(batch > 0 && hidden > 0, ARL_E_RANGE, "bad batch / hidden");
    GruBwdArgs a = {dh_or_null, dh_rec_or_null, dh_dir_or_null, saved, h_prev, dgx, dgh, dh_prev, batch, hidden,
                    dh_stride, saved_stride, hprev_stride, dgx_stride, dgh_stride};
    hipLaunchKernelGGL(gru_bwd_kernel, dim3(arl::stream_grid(batch * hidden, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return arl::check_launch("gru_bwd_kernel");
}

extern "C" int arl_rnn_cell_fwd(const float* gx, int64_t gx_stride, const float* gh, int64_t batch, int32_t hidden,
                                float* h_out, int64_t h_stride, void* stream) {
    ARL_REQUIRE(gx && gh && h_out, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(batch > 0 && hidden > 0, ARL_E_RANGE, "bad batch / hidden");
    RnnArgs a = {gx, gh, h_out, batch, hidden, gx_stride, h_stride};
    hipLaunchKernelGGL(rnn_fwd_kernel, dim3(arl::stream_grid(batch * hidden, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return arl::check_launch("rnn_fwd_kernel");
}

extern "C" int arl_rnn_cell_bwd(const float* dh_or_null, int64_t dh_stride, const float* dh_rec_or_null,
                                const float* h_out, int64_t h_stride, int64_t batch, int32_t hidden, float* dpre,
                                int64_t dpre_stride, void* stream) {
    ARL_REQUIRE(h_out && dpre, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(batch > 0 && hidden > 0, ARL_E_RANGE, "bad batch / hidden");
    RnnBwdArgs a = {dh_or_null, dh_rec_or_null, h_out, dpre, batch, hidden, dh_stride, h_stride, dpre_stride};
    hipLaunchKernelGGL(rnn_bwd_kernel, dim3(arl::stream_grid(batch * hidden, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return arl::check_launch("rnn_bwd_kernel");
}
