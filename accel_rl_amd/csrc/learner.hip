// Learner glue for gfx950: everything around the policy's conv / GEMM calls that
// the reference expresses as Theano graph nodes (and PyTorch would run as ~150
// tiny elementwise / reduction launches per minibatch), as a handful of fused,
// HBM-bound streaming kernels over channels-last activations.
//
//   arl_gather_scale_obs_nhwc  minibatch gather + u8 -> f32 * (1/255), NCHW u8 -> NHWC f32
//                              (optimizers/util.py:86-89 `s[idxs]`, policies/layers.py:22-41)
//   arl_bias_relu              x[M][C] = relu(x + b[c])          (Lasagne Conv2D/Dense b + rectify)
//   arl_relu_bwd_bias_grad     dy *= (y > 0); db[c] = sum_m dy   (their backward), deterministic
//   arl_pg_head_loss           policy/value heads + softmax + A2C / PPO / value / entropy losses
//                              and their gradients in one pass
//                              (policies/pg/networks/pg_cnn.py:70-86; algos/pg/aac_base.py:60-70;
//                               a2c.py:43-46; ppo.py:42-51; distributions/categorical.py:35-88)
//   arl_pg_head_wgrad          dW_head = dout^T h, db_head = colsum(dout)
//   arl_pg_head_infer          heads + softmax for action serving (atari_cnn_policy.py:67,109)
//
// All fp32 (the reference's floatX); compiled with -ffp-contract=off.

#include "arl_common.h"
#include "head_dev.h"
#include "dgrad_wt_dev.h"

namespace {

constexpr int HID_MAX = 1024;      // hidden width supported by the head kernels
constexpr int K_MAX = ARL_MAX_ACTIONS + 1;

// ---------------------------------------------------------------- gather NCHW u8 -> NHWC f32 (C = 4)
// one lane: 4 consecutive pixels of all 4 planes -> 4 float4 (64 B) stores
__global__ __launch_bounds__(256) void gather_nhwc4_kernel(const uint8_t* __restrict__ obs,
                                                           const int32_t* __restrict__ idx,
                                                           int64_t batch, int plane, float scale,
                                                           float* __restrict__ out) {
    const int q_per_row = plane >> 2;                       // 4-pixel groups per plane
    const int64_t total = batch * q_per_row;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t b = i / q_per_row;
        const int q = (int)(i - b * q_per_row);
        const int64_t src = (idx ? (int64_t)idx[b] : b) * 4 * plane;
        uint32_t w[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            w[c] = *reinterpret_cast<const uint32_t*>(obs + src + (int64_t)c * plane + (q << 2));
        // The 4 lanes of a quad hold 16 consecutive pixels (same image: plane % 16 == 0 keeps quads inside
        // one row of groups).  Transposed inside the quad, store p of lane j writes pixel 4p + j, so every
        // store instruction covers whole 64-byte runs instead of 16-byte pieces of four different ones.
        const int j = threadIdx.x & 3;
        float4* o = reinterpret_cast<float4*>(out) + (b * plane + ((q & ~3) << 2)) + j;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            uint32_t v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                v[c] = (uint32_t)__shfl((int)w[c], (threadIdx.x & ~3) | p, 64) >> (8 * j);
            o[4 * p] = make_float4((float)(v[0] & 255u) * scale, (float)(v[1] & 255u) * scale,
                                   (float)(v[2] & 255u) * scale, (float)(v[3] & 255u) * scale);
        }
    }
}

// ---------------------------------------------------------------- bias + relu, in place
__global__ __launch_bounds__(256) void bias_relu_kernel(float4* __restrict__ x,
                                                        const float4* __restrict__ bias,
                                                        int64_t total4, int c4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
        const float4 b = bias[i % c4];
        float4 v = x[i];
        v.x = fmaxf(v.x + b.x, 0.f); v.y = fmaxf(v.y + b.y, 0.f);
        v.z = fmaxf(v.z + b.z, 0.f); v.w = fmaxf(v.w + b.w, 0.f);
        x[i] = v;
    }
}

// ---------------------------------------------------------------- relu backward + bias grad
// thread -> (row lane rl, float4 column cg); fixed-order reductions => deterministic.
__global__ __launch_bounds__(256) void relu_bwd_bias_kernel(float4* __restrict__ dy,
                                                            const float4* __restrict__ y, int64_t m,
                                                            int c4, float4* __restrict__ partials) {
    __shared__ float4 lds[256];
    const int rows_per_iter = 256 / c4;
    const int cg = threadIdx.x % c4, rl = threadIdx.x / c4;
    float4 acc = make_float4(0, 0, 0, 0);
    if (rl < rows_per_iter) {
        for (int64_t row = (int64_t)blockIdx.x * rows_per_iter + rl; row < m;
             row += (int64_t)gridDim.x * rows_per_iter) {
            const int64_t i = row * c4 + cg;
            float4 g = dy[i];
            const float4 a = y[i];
            g.x = a.x > 0.f ? g.x : 0.f; g.y = a.y > 0.f ? g.y : 0.f;
            g.z = a.z > 0.f ? g.z : 0.f; g.w = a.w > 0.f ? g.w : 0.f;
            dy[i] = g;
            acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
        }
    }
    lds[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0) {
        float4 s = make_float4(0, 0, 0, 0);
        for (int r = 0; r < rows_per_iter; ++r) {
            const float4 v = lds[r * c4 + cg];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        partials[(int64_t)blockIdx.x * c4 + cg] = s;
    }
}

// out[c] = sum_g partials[g][c].  Block = 64 columns x 16 waves: wave w sums the rows
// g = w, w+16, ... (coalesced 256-B loads, independent accumulators), then the 16 wave
// sums are folded in a fixed order => deterministic and ~16x shorter dependency chains
// than one thread walking all partials.
__global__ __launch_bounds__(1024) void fold_partials_kernel(const float* __restrict__ partials,
                                                             int n_partials, int width,
                                                             float* __restrict__ out) {
    __shared__ float lds[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f;
    if (c < width) {
        int g = wave;
        for (; g + 16 < n_partials; g += 32) {
            s0 += partials[(int64_t)g * width + c];
            s1 += partials[(int64_t)(g + 16) * width + c];
        }
        if (g < n_partials) s0 += partials[(int64_t)g * width + c];
    }
    lds[wave][lane] = s0 + s1;
    __syncthreads();
    if (wave == 0 && c < width) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) s += lds[w][lane];
        out[c] = s;
    }
}

// ---------------------------------------------------------------- heads

struct HeadLossArgs {
    const float* h;          // [B][hid] post-relu hidden activations
    const float* w_head;     // [K][hid], rows 0..A-1 = pi, row A = value
    const float* b_head;     // [K]
    const uint8_t* actions;  // [n_rows]
    const float* adv;        // [n_rows]
    const float* ret;        // [n_rows]
    const float* old_prob;   // [n_rows][A]
    const int8_t* valids;    // [n_rows] or null
    const int32_t* idx;      // [B] rows of this minibatch, or null (identity)
    const float* lr_mult;    // device scalar (PPO clip anneals with it, ppo.py:46)
    const float* inv_count;  // device scalar 1/sum(valids) or null -> 1/B
    float* dout;             // [B][K] gradient wrt (logits, value)
    float* dh;               // [B][hid] gradient wrt h (before the relu mask)
    float* loss_partials;    // [gridDim][4] = pi, v, ent, their sum
    float* wpart;            // [gridDim][K][hid] this workgroup's share of dW_head = sum_b dout[b] x h[b], or null
    float* bpart;            // [gridDim][Kp]     ... of db_head = sum_b dout[b]                (with wpart)
    int batch, hid, n_act, kind, tie_rule;
    float clip_param, v_coeff, ent_coeff;
    int mask_dh;             // dh *= (h > 0): h is a rectifier's output and the caller wants the gradient before it
    int head_blocks;         // workgroups of the head itself (the launch may carry more: see head_kernel)
};

// one wave per row (looping); lanes split the hidden dimension.  Every per-action
// array is indexed with compile-time indices only (fully unrolled loops with
// `k < n_act` guards) so that it lives in VGPRs, not scratch.
// HVT = hid / 64 when the hidden width is a multiple of 64 (no per-lane guards: a guarded element costs an
// exec-mask save / branch / restore, and the generic kernel is mostly those), 0 = any width.
// wt (TRAIN): the workgroups behind the head's own grid (blockIdx.x >= a.head_blocks) write the data gradients' k-contiguous
// weight copies (dgrad_wt_dev.h) -- the backward pass that follows reads them, and this launch is where a minibatch's
// parameters are final and nothing else needs the CUs (one launch less per minibatch).
template <bool TRAIN, int HVT = 0>
__global__ __launch_bounds__(256) void head_kernel(HeadLossArgs a, float* __restrict__ prob_out,
                                                   float* __restrict__ value_out, const arlw::DgradWtArgs wt) {
    if (TRAIN && (int)blockIdx.x >= a.head_blocks) {
        arlw::dgrad_wt_block(wt, (int)blockIdx.x - a.head_blocks, (int)threadIdx.x);
        return;
    }
    constexpr int HV = HVT ? HVT : HID_MAX / 64;
    extern __shared__ __attribute__((aligned(16))) float s_w[];     // [K][hid] (+ [4][K][hid] head-gradient slices)
    __shared__ float s_loss[4][4];
    __shared__ float s_db[4][64];
    const int A = a.n_act, K = A + 1, hid = a.hid, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // A row's scalars hang off a three-deep chain of dependent loads (idx -> action -> old probability): they
    // are fetched one row ahead -- the first row's before the weights are even staged -- so the chain's
    // latency runs under the dot products instead of after them.
    struct RowMeta { int64_t row; int act; float adv, ret, valid, old_pa; };
    auto load_meta = [&](int b) {
        RowMeta m = {b, 0, 0.f, 0.f, 1.f, 0.f};
        if (TRAIN && b < a.batch) {
            m.row = a.idx ? (int64_t)a.idx[b] : b;
            m.act = a.actions[m.row];
            m.adv = a.adv[m.row]; m.ret = a.ret[m.row];
            m.valid = a.valids ? (a.valids[m.row] != 0 ? 1.f : 0.f) : 1.f;
            if (a.kind == 1) m.old_pa = a.old_prob[m.row * A + m.act];
        }
        return m;
    };
    const int waves_total = ((TRAIN ? a.head_blocks : (int)gridDim.x) * blockDim.x) >> 6;
    RowMeta meta = load_meta(blockIdx.x * (blockDim.x >> 6) + wave);
    if (((K * hid) & 3) || (reinterpret_cast<uintptr_t>(a.w_head) & 15)) {
        for (int i = threadIdx.x; i < K * hid; i += blockDim.x) s_w[i] = a.w_head[i];
    } else {    // stage W: independent b128 loads, four in flight per thread
        const int n4 = (K * hid) >> 2;
        const float4* src = reinterpret_cast<const float4*>(a.w_head);
        float4* dst = reinterpret_cast<float4*>(s_w);
        for (int i0 = threadIdx.x; i0 < n4; i0 += 4 * 256) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * 256 < n4) v[u] = src[i0 + u * 256];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * 256 < n4) dst[i0 + u * 256] = v[u];
        }
    }
    __syncthreads();
    const float TINY = 1e-8f;                                        // categorical.py:6
    const float inv_n = TRAIN ? (a.inv_count ? a.inv_count[0] : 1.f / (float)a.batch) : 0.f;
    const float clip = TRAIN ? a.clip_param * a.lr_mult[0] : 0.f;
    float l_pi = 0.f, l_v = 0.f, l_ent = 0.f;
    // The head's own weight / bias gradient rides along (no second pass over dout and h): every wave adds its rows'
    // outer products dout[b] x h[b] into its LDS slice, the workgroup sums the four slices in wave order and leaves
    // ONE partial per workgroup for arl_fold_many (fixed order => deterministic).
    const bool fused_wgrad = TRAIN && a.wpart != nullptr;            // uniform
    float* s_dw = s_w + K * hid + wave * K * hid;
    float db_lane = 0.f;
    if (fused_wgrad) {
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int j = 0; j < HV; ++j)
                if (HVT || lane + 64 * j < hid) s_dw[k * hid + lane + 64 * j] = 0.f;
    }
    const int waves = waves_total;
    for (int b = blockIdx.x * (blockDim.x >> 6) + wave; b < a.batch; b += waves) {
        const RowMeta cur = meta;
        meta = load_meta(b + waves);                                    // next row's chain starts now
        const float* hrow = a.h + (int64_t)b * hid;
        float hv[HV];
#pragma unroll
        for (int j = 0; j < HV; ++j) hv[j] = (HVT || lane + 64 * j < hid) ? hrow[lane + 64 * j] : 0.f;
        // Lane k owns action k (lane A the value): the transcendental work is done once per action, not
        // once per lane; sums over actions stay sequential in k (readlane), i.e. in a fixed order.  The loops
        // over k are real loops (K is uniform): unrolled to the 19-action maximum they were mostly branches.
        float v = 0.f, mx = -3.0e38f, mine = 0.f;
        for (int k = 0; k < K; ++k) {
            float sdot = 0.f;
#pragma unroll
            for (int j = 0; j < HV; ++j)
                if (HVT || lane + 64 * j < hid) sdot += hv[j] * s_w[k * hid + lane + 64 * j];
            const float o = wave_sum_f(sdot) + a.b_head[k];
            if (k < A) mx = fmaxf(mx, o); else v = o;
            mine = (lane == k) ? o : mine;
        }
        const bool is_act = lane < A;
        const float ex = is_act ? expf(mine - mx) : 0.f;
        float z = 0.f;
        for (int k = 0; k < A; ++k) z += readlane_f(ex, k);
        const float pk = ex / z;
        if (!TRAIN) {
            if (is_act) prob_out[(int64_t)b * A + lane] = pk;
            if (lane == 0) value_out[b] = v;
            continue;
        }
        const float w = cur.valid * inv_n;                              // valids_mean
        const int act = cur.act;
        const float adv = cur.adv, ret = cur.ret;
        const float pa = __shfl(pk, act, 64);
        // ---- d loss / d p_k : entropy term for every action (categorical.py:76-78)
        const float lg = is_act ? logf(pk + TINY) : 0.f;
        const float ent_k = pk * lg;
        float gk = is_act ? a.ent_coeff * w * (lg + pk / (pk + TINY)) : 0.f;     // d(-c_e * ent)/dp_k
        float ent = 0.f;
        for (int k = 0; k < A; ++k) ent -= readlane_f(ent_k, k);
        float pi_term, g_act;
        if (a.kind == 1) {                                               // PPO, ppo.py:42-51
            const float old_pa = cur.old_pa;
            const float ratio = (pa + TINY) / (old_pa + TINY);           // categorical.py:66-70
            const float lo = 1.f - clip, hi = 1.f + clip;
            const float rc = fminf(fmaxf(ratio, lo), hi);
            const float s1 = ratio * adv, s2 = rc * adv;
            pi_term = fminf(s1, s2);
            const bool in_range = (ratio >= lo) && (ratio <= hi);
            float g_ratio;
            if (a.tie_rule == ARL_PPO_TIE_THEANO) {
                // the reference's graph as Theano >= 0.8 differentiates it: Minimum.L_op, e = eq(min, x), gives e g to
                // its FIRST argument (s1 here: T.minimum(surr_1, surr_2)) and (1 - e) g to the second -- a tie goes to
                // the first alone --, Clip.L_op passes g for lo <= r <= hi
                g_ratio = (pi_term == s1) ? adv : (in_range ? adv : 0.f);
            } else if (a.tie_rule == ARL_PPO_TIE_BOTH) {
                // Theano <= 0.7: eq(min, x) g to EVERY argument equal to the minimum -- inside the range s1 and s2 are the
                // same number and the sample counts twice
                g_ratio = ((pi_term == s1) ? adv : 0.f) + ((pi_term == s2 && in_range) ? adv : 0.f);
            } else {
                g_ratio = in_range ? adv : (s1 < s2 ? adv : 0.f);
            }
            g_act = -w * g_ratio / (old_pa + TINY);
        } else {                                                         // A2C, a2c.py:43-46
            pi_term = logf(pa + TINY) * adv;
            g_act = -w * adv / (pa + TINY);
        }
        const float dv = 2.f * a.v_coeff * w * (v - ret);                // aac_base.py:60-61
        gk += (lane == act) ? g_act : 0.f;
        const float gp = gk * pk;
        float dot = 0.f;
        for (int k = 0; k < A; ++k) dot += readlane_f(gp, k);
        const float dlk = is_act ? pk * (gk - dot) : (lane == A ? dv : 0.f);      // softmax backward | value
        if (lane < K) a.dout[(int64_t)b * K + lane] = dlk;
        if (lane == 0) {
            l_pi += -w * pi_term;
            l_v += a.v_coeff * w * (v - ret) * (v - ret);
            l_ent += -a.ent_coeff * w * ent;
        }
        // dh[b][c] = sum_k dl_k W[k][c], k in order
        float sdh[HV];
#pragma unroll
        for (int j = 0; j < HV; ++j) sdh[j] = 0.f;
        for (int k = 0; k < K; ++k) {
            const float dl_k = readlane_f(dlk, k);
#pragma unroll
            for (int j = 0; j < HV; ++j)
                if (HVT || lane + 64 * j < hid) sdh[j] += dl_k * s_w[k * hid + lane + 64 * j];
            if (fused_wgrad) {
#pragma unroll
                for (int j = 0; j < HV; ++j)
                    if (HVT || lane + 64 * j < hid) s_dw[k * hid + lane + 64 * j] += dl_k * hv[j];
            }
        }
        if (lane < K) db_lane += dlk;
        float* dhrow = a.dh + (int64_t)b * hid;
#pragma unroll
        for (int j = 0; j < HV; ++j) {
            const int c = lane + 64 * j;
            if (HVT || c < hid) dhrow[c] = (a.mask_dh && !(hv[j] > 0.f)) ? 0.f : sdh[j];
        }
    }
    if (TRAIN) {
        if (lane == 0) { s_loss[wave][0] = l_pi; s_loss[wave][1] = l_v; s_loss[wave][2] = l_ent; s_loss[wave][3] = (l_pi + l_v) + l_ent; }
        __syncthreads();
        if (threadIdx.x < 4) {
            float sl = 0.f;
            for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) sl += s_loss[wv][threadIdx.x];
            a.loss_partials[blockIdx.x * 4 + threadIdx.x] = sl;
        }
        if (fused_wgrad) {                                               // (the barrier above covers the slices too)
            s_db[wave][lane] = db_lane;
            const float* s0 = s_w + K * hid;
            float* out = a.wpart + (int64_t)blockIdx.x * K * hid;
            for (int i = threadIdx.x; i < K * hid; i += blockDim.x)
                out[i] = ((s0[i] + s0[K * hid + i]) + s0[2 * K * hid + i]) + s0[3 * K * hid + i];
            __syncthreads();
            const int Kp = (K + 3) & ~3;
            if (threadIdx.x < Kp)
                a.bpart[(int64_t)blockIdx.x * Kp + threadIdx.x] =
                    threadIdx.x < K ? ((s_db[0][threadIdx.x] + s_db[1][threadIdx.x]) + s_db[2][threadIdx.x]) + s_db[3][threadIdx.x] : 0.f;
        }
    }
}

// Partial head weight gradient: part[rs][k][c] = sum_{b in row split rs} dout[b][k] h[b][c],
// k == K holds the bias gradient (h := 1).  grid = (hid/64 column tiles, row splits); the
// 4 waves of a block take rows b = wave, wave+4, ... of the split; dout rows sit in LDS.
// Folded over row splits by arl_fold_many (fixed order) => deterministic.
constexpr int WG_SPLITS = 16;
// the head kernel sums the head's weight gradient itself while its four LDS slices [4][K][hid] (next to the staged
// weights [K][hid]) fit 64 KB: K * hid <= 3 072, e.g. 5 x 512, 19 x 128
constexpr int FUSED_WGRAD_MAX = 3072;

__global__ __launch_bounds__(256) void head_wgrad_kernel(const float* __restrict__ dout,
                                                         const float* __restrict__ h, int batch,
                                                         int hid, int K, int Kp, float* __restrict__ part,
                                                         float* __restrict__ part_b) {
    __shared__ float lds[4][K_MAX][64];
    extern __shared__ __attribute__((aligned(16))) float s_dout[];   // [rows_in_split][K]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int rows_per = (batch + WG_SPLITS - 1) / WG_SPLITS;
    const int b0 = blockIdx.y * rows_per;
    const int b1 = (b0 + rows_per < batch) ? b0 + rows_per : batch;
    const int n_rows = b1 > b0 ? b1 - b0 : 0;
    for (int i = threadIdx.x; i < n_rows * K; i += blockDim.x) s_dout[i] = dout[(int64_t)b0 * K + i];
    __syncthreads();
    float acc[K_MAX];
#pragma unroll
    for (int k = 0; k < K_MAX; ++k) acc[k] = 0.f;
    if (c < hid) {
        for (int r = wave; r < n_rows; r += 4) {
            const float hv = h[(int64_t)(b0 + r) * hid + c];
#pragma unroll
            for (int k = 0; k < K_MAX; ++k)
                if (k < K) acc[k] += s_dout[r * K + k] * hv;
        }
    }
#pragma unroll
    for (int k = 0; k < K_MAX; ++k) lds[wave][k][lane] = acc[k];
    __syncthreads();
    if (wave == 0 && c < hid) {
        for (int k = 0; k < K; ++k)
            part[((int64_t)blockIdx.y * K + k) * hid + c] =
                ((lds[0][k][lane] + lds[1][k][lane]) + lds[2][k][lane]) + lds[3][k][lane];
    }
    // bias gradient: part_b[split][Kp], column c < K holds sum_b dout[b][c] (Kp = K rounded up to 4, padding zero)
    if (blockIdx.x == 0 && wave == 1 && lane < Kp) {
        float sb = 0.f;
        if (lane < K)
            for (int r = 0; r < n_rows; ++r) sb += s_dout[r * K + lane];
        part_b[(int64_t)blockIdx.y * Kp + lane] = sb;
    }
}

}  // namespace

extern "C" int arl_gather_scale_obs_nhwc(const uint8_t* obs, const int32_t* idx_or_null,
                                         int64_t batch, int32_t channels, int32_t plane_bytes,
                                         float scale, float* out, void* stream) {
    ARL_REQUIRE(obs && out, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(batch >= 0 && plane_bytes > 0, ARL_E_ARG, "bad batch/plane");
    ARL_REQUIRE(channels == 4, ARL_E_RANGE, "NHWC gather is specialised for 4 stacked frames");
    ARL_REQUIRE((plane_bytes & 15) == 0, ARL_E_RANGE, "plane_bytes must be a multiple of 16");
    ARL_REQUIRE(arl::aligned4(obs) && arl::aligned16(out), ARL_E_ALIGN, "obs 4-byte, out 16-byte aligned");
    if (batch == 0) return 0;
    hipLaunchKernelGGL(gather_nhwc4_kernel, dim3(arl::stream_grid(batch * (plane_bytes >> 2), 256)),
                       dim3(256), 0, (hipStream_t)stream, obs, idx_or_null, batch, (int)plane_bytes,
                       scale, out);
    return arl::check_launch("gather_nhwc4_kernel");
}

extern "C" int arl_bias_relu(float* x, const float* bias, int64_t rows, int32_t channels, void* stream) {
    ARL_REQUIRE(x && bias, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(rows >= 0 && channels > 0 && (channels & 3) == 0, ARL_E_RANGE, "channels must be a multiple of 4");
    ARL_REQUIRE(arl::aligned16(x) && arl::aligned16(bias), ARL_E_ALIGN, "x/bias must be 16-byte aligned");
    if (rows == 0) return 0;
    const int64_t total4 = rows * (channels >> 2);
    hipLaunchKernelGGL(bias_relu_kernel, dim3(arl::stream_grid(total4, 256)), dim3(256), 0,
                       (hipStream_t)stream, (float4*)x, (const float4*)bias, total4, (int)(channels >> 2));
    return arl::check_launch("bias_relu_kernel");
}

extern "C" int64_t arl_relu_bwd_workspace_bytes(void) { return (int64_t)256 * HID_MAX * sizeof(float); }
extern "C" int64_t arl_pg_head_workspace_bytes(void) {
    // loss partials [256][4]; head-gradient partials: [WG_SPLITS][K][hid] (separate kernel) or [<= 256][K][hid] with
    // K * hid <= FUSED_WGRAD_MAX (summed inside the head kernel); bias partials [<= 256][Kp]
    return (int64_t)(256 * 4 + 256 * FUSED_WGRAD_MAX + WG_SPLITS * (K_MAX + 1) * HID_MAX + 256 * (K_MAX + 4)) * sizeof(float);
}

static int relu_bwd_bias_launch(float* dy, const float* y, int64_t rows, int32_t channels, float* dbias,
                                void* workspace, int* n_partials, void* stream) {
    ARL_REQUIRE(dy && y && dbias && workspace, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(rows > 0 && channels > 0 && (channels & 3) == 0 && channels <= HID_MAX, ARL_E_RANGE,
                "channels must be a multiple of 4 and <= 1024");
    ARL_REQUIRE(arl::aligned16(dy) && arl::aligned16(y) && arl::aligned16(workspace), ARL_E_ALIGN, "16-byte alignment");
    const int c4 = channels >> 2, rows_per_iter = 256 / c4;
    int64_t grid = (rows + rows_per_iter - 1) / rows_per_iter;
    if (grid > 256) grid = 256;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(relu_bwd_bias_kernel, dim3((unsigned)grid), dim3(256), 0, s, (float4*)dy,
                       (const float4*)y, rows, c4, (float4*)workspace);
    *n_partials = (int)grid;
    return arl::check_launch("relu_bwd_bias_kernel");
}

extern "C" int arl_relu_bwd_bias_grad(float* dy, const float* y, int64_t rows, int32_t channels,
                                      float* dbias, void* workspace, void* stream) {
    int grid = 0;
    int rc = relu_bwd_bias_launch(dy, y, rows, channels, dbias, workspace, &grid, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(fold_partials_kernel, dim3((channels + 63) / 64), dim3(1024), 0, (hipStream_t)stream,
                       (const float*)workspace, grid, (int)channels, dbias);
    return arl::check_launch("fold_partials_kernel");
}

extern "C" int arl_relu_bwd_bias_parts(float* dy, const float* y, int64_t rows, int32_t channels, float* dbias,
                                       void* workspace, arl_fold_item* item, void* stream) {
    ARL_REQUIRE(item, ARL_E_ARG, "null pointer");
    int grid = 0;
    int rc = relu_bwd_bias_launch(dy, y, rows, channels, dbias, workspace, &grid, stream);
    item->part = (const float*)workspace; item->out = dbias; item->total = channels; item->splits = grid; item->valid = 0;
    return rc;
}

static int check_head(int64_t batch, int32_t hid, int32_t n_act) {
    if (batch <= 0 || hid <= 0 || hid > HID_MAX || n_act <= 0 || n_act > ARL_MAX_ACTIONS) {
        arl::set_error("head: bad batch/hidden/n_actions");
        return ARL_E_RANGE;
    }
    return 0;
}

extern "C" int arl_pg_head_infer(const float* h, const float* w_head, const float* b_head,
                                 int64_t batch, int32_t hid, int32_t n_actions, float* prob,
                                 float* value, void* stream) {
    ARL_REQUIRE(h && w_head && b_head && prob && value, ARL_E_ARG, "null pointer");
    int rc = check_head(batch, hid, n_actions);
    if (rc) return rc;
    HeadLossArgs a = {};
    a.h = h; a.w_head = w_head; a.b_head = b_head; a.batch = (int)batch; a.hid = hid; a.n_act = n_actions;
    const int grid = (int)((batch + 3) / 4 < 1024 ? (batch + 3) / 4 : 1024);
    const size_t lds = (size_t)(n_actions + 1) * hid * 4;
#define ARL_HEAD_INFER(HVT_) hipLaunchKernelGGL((head_kernel<false, HVT_>), dim3(grid), dim3(256), lds, (hipStream_t)stream, a, prob, value, arlw::DgradWtArgs{})
    if (hid == 512) ARL_HEAD_INFER(8);
    else if (hid == 256) ARL_HEAD_INFER(4);
    else if (hid == 1024) ARL_HEAD_INFER(16);
    else if (hid == 64) ARL_HEAD_INFER(1);
    else ARL_HEAD_INFER(0);
#undef ARL_HEAD_INFER
    return arl::check_launch("head_kernel<infer>");
}

extern "C" int arl_pg_head_loss_parts(const float* h, const float* w_head, const float* b_head,
                                const uint8_t* actions, const float* advantages, const float* returns,
                                const float* old_prob, const int8_t* valids_or_null,
                                const int32_t* idx_or_null, const float* lr_mult,
                                const float* inv_count_or_null, int64_t batch, int32_t hid,
                                int32_t n_actions, int32_t kind, int32_t tie_rule, float clip_param, float v_loss_coeff,
                                float ent_loss_coeff, int32_t relu_mask_dh, float* dout, float* dh,
                                float* dw_head, float* db_head, float* loss4, void* workspace, arl_fold_item* items3,
                                      const arl_dgrad_wt* wt_items_or_null, int32_t n_wt, void* stream) {
    ARL_REQUIRE(h && w_head && b_head && actions && advantages && returns && lr_mult && dout && dh &&
                    dw_head && db_head && loss4 && workspace && items3, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(kind == 0 || (kind == 1 && old_prob), ARL_E_ARG, "kind must be 0 (A2C) or 1 (PPO, needs old_prob)");
    ARL_REQUIRE(tie_rule == ARL_PPO_TIE_THEANO || tie_rule == ARL_PPO_TIE_MATH || tie_rule == ARL_PPO_TIE_BOTH, ARL_E_ARG,
                "tie_rule must be ARL_PPO_TIE_THEANO, ARL_PPO_TIE_MATH or ARL_PPO_TIE_BOTH");
    int rc = check_head(batch, hid, n_actions);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    HeadLossArgs a = {};
    a.h = h; a.w_head = w_head; a.b_head = b_head; a.actions = actions; a.adv = advantages;
    a.ret = returns; a.old_prob = old_prob; a.valids = valids_or_null; a.idx = idx_or_null;
    a.lr_mult = lr_mult; a.inv_count = inv_count_or_null; a.dout = dout; a.dh = dh;
    a.loss_partials = (float*)workspace;
    a.batch = (int)batch; a.hid = hid; a.n_act = n_actions; a.kind = kind; a.tie_rule = tie_rule;
    a.clip_param = clip_param; a.v_coeff = v_loss_coeff; a.ent_coeff = ent_loss_coeff; a.mask_dh = relu_mask_dh;
    const int grid = (int)((batch + 3) / 4 < 256 ? (batch + 3) / 4 : 256);
    const int K = n_actions + 1;
    const int Kp = (K + 3) & ~3;
    float* ws = (float*)workspace;
    float* part = ws + 256 * 4;
    const bool fused = K * hid <= FUSED_WGRAD_MAX;
    float* part_b = part + (fused ? (int64_t)grid : (int64_t)WG_SPLITS) * K * hid;
    if (fused) { a.wpart = part; a.bpart = part_b; }
    const size_t head_lds = (size_t)(fused ? 5 : 1) * K * hid * 4;
    arlw::DgradWtArgs wt = {};
    int wt_blocks = 0;
    if (wt_items_or_null && n_wt > 0) {
        rc = arlw::dgrad_wt_plan(wt_items_or_null, n_wt, &wt, &wt_blocks);
        if (rc) return rc;
    }
    a.head_blocks = grid;
#define ARL_HEAD_TRAIN(HVT_) hipLaunchKernelGGL((head_kernel<true, HVT_>), dim3(grid + wt_blocks), dim3(256), head_lds, s, a, (float*)nullptr, (float*)nullptr, wt)
    if (hid == 512) ARL_HEAD_TRAIN(8);
    else if (hid == 256) ARL_HEAD_TRAIN(4);
    else if (hid == 1024) ARL_HEAD_TRAIN(16);
    else if (hid == 64) ARL_HEAD_TRAIN(1);
    else ARL_HEAD_TRAIN(0);
#undef ARL_HEAD_TRAIN
    rc = arl::check_launch("head_kernel<train>");
    if (rc) return rc;
    // head weight / bias gradient: row-split partials [WG_SPLITS][K][hid] and [WG_SPLITS][Kp]; their folds and the
    // fold of the per-workgroup loss partials [grid][4] are left to arl_fold_many (one launch per backward pass)
    if (!fused) {
        const int rows_per = ((int)batch + WG_SPLITS - 1) / WG_SPLITS;
        hipLaunchKernelGGL(head_wgrad_kernel, dim3((hid + 63) / 64, WG_SPLITS), dim3(256),
                           (size_t)rows_per * K * 4, s, dout, h, (int)batch, (int)hid, K, Kp, part, part_b);
        rc = arl::check_launch("head_wgrad_kernel");
        if (rc) return rc;
    }
    const int n_parts = fused ? grid : WG_SPLITS;
    items3[0].part = part; items3[0].out = dw_head; items3[0].total = (int64_t)K * hid; items3[0].splits = n_parts;
    items3[1].part = part_b; items3[1].out = db_head; items3[1].total = Kp; items3[1].splits = n_parts;
    items3[2].part = ws; items3[2].out = loss4; items3[2].total = 4; items3[2].splits = grid;
    items3[0].valid = items3[2].valid = 0;
    items3[1].valid = K;                                // db_head holds K floats, the partials are padded to Kp
    return 0;
}

extern "C" int arl_pg_head_loss(const float* h, const float* w_head, const float* b_head,
                                const uint8_t* actions, const float* advantages, const float* returns,
                                const float* old_prob, const int8_t* valids_or_null,
                                const int32_t* idx_or_null, const float* lr_mult,
                                const float* inv_count_or_null, int64_t batch, int32_t hid,
                                int32_t n_actions, int32_t kind, int32_t tie_rule, float clip_param, float v_loss_coeff,
                                float ent_loss_coeff, int32_t relu_mask_dh, float* dout, float* dh,
                                float* dw_head, float* db_head, float* loss4, void* workspace, void* stream) {
    arl_fold_item items[3];
    int rc = arl_pg_head_loss_parts(h, w_head, b_head, actions, advantages, returns, old_prob, valids_or_null,
                                    idx_or_null, lr_mult, inv_count_or_null, batch, hid, n_actions, kind, tie_rule, clip_param,
                                    v_loss_coeff, ent_loss_coeff, relu_mask_dh, dout, dh, dw_head, db_head, loss4,
                                    workspace, items, nullptr, 0, stream);
    if (rc) return rc;
    return arl_fold_many(items, 3, stream);
}
