// LSTM cell for the recurrent policies: the elementwise part of
// FastLstmLayer.step, accel_rl/policies/layers.py:331-346 -- gate order f, i, c~, o ("fico"),
// sigmoid gates, tanh cell/output nonlinearity -- and its backward.  The two matrix products
// (x W_x + b, h_prev W_h) are dense fp32-MFMA calls (csrc/mfma_conv.hip); these kernels stream
// the gate pre-activations once.  Rows may be strided (row_stride elements between consecutive
// rows) so that one time slice t of a [trajectory][time] batch is addressed in place.
//   fwd:  f,i,o = sigmoid(.), g = tanh(.); c = f c_prev + i g; h = o tanh(c)
//   bwd:  given dh (all sources summed) and dc_next: do = dh tanh(c); dc = dc_next + dh o (1 - tanh(c)^2);
//         df = dc c_prev; di = dc g; dg = dc i; dc_prev = dc f; pre-activation grads via s(1-s), 1-g^2

#include "arl_common.h"

namespace {

struct LstmArgs {
    const float* gx;        // [B][4H] (+ strides): x W_x + b
    const float* gh;        // [B][4H] contiguous: h_prev W_h, or null
    const float* c_prev;    // [B][H]
    float* h_out;           // [B][H]
    float* c_out;           // [B][H]
    float* gates;           // [B][4H] activated gates saved for the backward pass, or null
    int64_t batch;
    int hidden;
    int64_t gx_stride, cprev_stride, h_stride, c_stride, gates_stride;   // elements between rows
};

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void lstm_fwd_kernel(const LstmArgs a) {
    const int H = a.hidden;
    const int64_t total = a.batch * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / H;
        const int j = (int)(i - b * H);
        const float* gx = a.gx + b * a.gx_stride;
        float pf = gx[j], pi = gx[H + j], pg = gx[2 * H + j], po = gx[3 * H + j];
        if (a.gh) {
            const float* gh = a.gh + b * 4 * H;
            pf += gh[j]; pi += gh[H + j]; pg += gh[2 * H + j]; po += gh[3 * H + j];
        }
        const float f = sigmoidf(pf), ig = sigmoidf(pi), g = tanhf(pg), o = sigmoidf(po);
        const float c = f * a.c_prev[b * a.cprev_stride + j] + ig * g;
        a.c_out[b * a.c_stride + j] = c;
        a.h_out[b * a.h_stride + j] = o * tanhf(c);
        if (a.gates) {
            float* gs = a.gates + b * a.gates_stride;
            gs[j] = f; gs[H + j] = ig; gs[2 * H + j] = g; gs[3 * H + j] = o;
        }
    }
}

struct LstmBwdArgs {
    const float* dh;        // [B][H] gradient wrt h_out from the layers above (strided), or null
    const float* dh_rec;    // [B][H] contiguous: gradient wrt h_out from the next time step, or null
    const float* dc_next;   // [B][H] contiguous, or null
    const float* gates;     // [B][4H] activated gates (strided)
    const float* c_prev;    // [B][H] (strided)
    const float* c_out;     // [B][H] (strided)
    float* dgates;          // [B][4H] pre-activation gradients (strided)
    float* dc_prev;         // [B][H] contiguous
    int64_t batch;
    int hidden;
    int64_t dh_stride, gates_stride, cprev_stride, c_stride, dgates_stride;
};

__global__ __launch_bounds__(256) void lstm_bwd_kernel(const LstmBwdArgs a) {
    const int H = a.hidden;
    const int64_t total = a.batch * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / H;
        const int j = (int)(i - b * H);
        const float* gs = a.gates + b * a.gates_stride;
        const float f = gs[j], ig = gs[H + j], g = gs[2 * H + j], o = gs[3 * H + j];
        float dh = 0.f;
        if (a.dh) dh += a.dh[b * a.dh_stride + j];
        if (a.dh_rec) dh += a.dh_rec[i];
        const float tc = tanhf(a.c_out[b * a.c_stride + j]);
        float dc = dh * o * (1.f - tc * tc);
        if (a.dc_next) dc += a.dc_next[i];
        float* dg = a.dgates + b * a.dgates_stride;
        dg[j] = dc * a.c_prev[b * a.cprev_stride + j] * f * (1.f - f);
        dg[H + j] = dc * g * ig * (1.f - ig);
        dg[2 * H + j] = dc * ig * (1.f - g * g);
        dg[3 * H + j] = dh * tc * o * (1.f - o);
        a.dc_prev[i] = dc * f;
    }
}

}  // namespace

extern "C" int arl_lstm_cell_fwd(const float* gx, int64_t gx_stride, const float* gh_or_null, const float* c_prev,
                                 int64_t cprev_stride, int64_t batch, int32_t hidden, float* h_out, int64_t h_stride,
                                 float* c_out, int64_t c_stride, float* gates_or_null, int64_t gates_stride,
                                 void* stream) {
    ARL_REQUIRE(gx && c_prev && h_out && c_out, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(batch > 0 && hidden > 0, ARL_E_RANGE, "bad batch / hidden");
    LstmArgs a = {gx, gh_or_null, c_prev, h_out, c_out, gates_or_null, batch, hidden,
                  gx_stride, cprev_stride, h_stride, c_stride, gates_stride};
    hipLaunchKernelGGL(lstm_fwd_kernel, dim3(arl::stream_grid(batch * hidden, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return arl::check_launch("lstm_fwd_kernel");
}

extern "C" int arl_lstm_cell_bwd(const float* dh_or_null, int64_t dh_stride, const float* dh_rec_or_null,
                                 const float* dc_next_or_null, const float* gates, int64_t gates_stride,
                                 const float* c_prev, int64_t cprev_stride, const float* c_out, int64_t c_stride,
                                 int64_t batch, int32_t hidden, float* dgates, int64_t dgates_stride,
                                 float* dc_prev, void* stream) {
    ARL_REQUIRE(gates && c_prev && c_out && dgates && dc_prev, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(batch > 0 && hidden > 0, ARL_E_RANGE, "bad batch / hidden");
    LstmBwdArgs a = {dh_or_null, dh_rec_or_null, dc_next_or_null, gates, c_prev, c_out, dgates, dc_prev, batch, hidden,
                     dh_stride, gates_stride, cprev_stride, c_stride, dgates_stride};
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3(arl::stream_grid(batch * hidden, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return arl::check_launch("lstm_bwd_kernel");
}
