// Categorical DQN ("C51") output stage for gfx950: everything after the network's last dense
// layer, which the reference expresses as Theano graph nodes.
//
//   arl_catdqn_act    per-action softmax over atoms, Q = sum_i p_i z_i, greedy action (first
//                     maximum, as T.argmax), epsilon-greedy override, served as a one-hot row so
//                     that the sampler's categorical kernel picks exactly that action
//                     (policies/dqn/atari_cat_dqn_policy.py:84-126, catdqn_cnn.py:94-99)
//   arl_catdqn_loss   distributional Bellman target projected on the fixed support, cross-entropy
//                     loss, KL priorities, and d loss / d logits in one pass
//                     (algos/dqn/cat_dqn.py:40-109)
//
//   arl_dqn_act       plain Q-learning twin of arl_catdqn_act: greedy action = first maximum of the Q row
//                     (policies/dqn/atari_dqn_policy.py:61-63,118-130)
//   arl_dqn_loss      one-step / n-step Q-learning target (max or double-DQN selection), squared or
//                     Huber loss, clipped |TD error| priorities, d loss / d Q (algos/dqn/dqn.py:137-172)
//
// Layout: logits f32[batch][n_actions (+ 1 value row when dueling)][atom_stride], atom_stride = n_atoms rounded up to a
// multiple of 4 (the dense layer producing them is an MFMA kernel with 16-byte rows); the padding
// columns are ignored on input and receive zero gradient.  Lane i owns atom i; one wave per sample (action
// kernel) / one workgroup per sample (loss kernel); n_atoms <= 64.  fp32, compiled with -ffp-contract=off.

#include "arl_common.h"
#include "dgrad_wt_dev.h"
#include <type_traits>

namespace {

__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = fmaxf(x, __shfl_xor(x, off, 64));
    return x;
}
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// softmax over the atoms of one action; lane i holds logit x (lanes >= n_atoms: anything) and returns p_i (0 there)
__device__ __forceinline__ float atom_softmax(float x, int lane, int n_atoms) {
    x = lane < n_atoms ? x : -3.0e38f;
    const float m = wave_max(x);
    const float e = lane < n_atoms ? expf(x - m) : 0.f;
    return e / wave_sum(e);
}

// Where a sample's logit block [A (+ 1)][stride] is read from: the finished logits, or (Parts) the output layer's split
// partial sums as arl_conv2d_fwd_parts left them -- logit = fold_splits_kernel's sum of the partials + bias, operation for
// operation (its 16 lanes-of-a-group partial sums s_g = 0 + p_g + p_{g+16} + ..., then s_0 + s_1 + ... + s_15, then the bias),
// so that the loss kernel can take the place of the fold launch without moving a bit.
struct Plain {
    const float* p;
    __device__ __forceinline__ float operator()(int off) const { return p[off]; }
};
template <int NS>                  // NS: compile-time number of splits (1, 2, 4, 8, 16), 0 = run-time (<= 127)
struct Parts {
    const float* p;             // this sample's block inside split 0
    const float* bias;          // [A (+ 1)][stride] (the output layer's bias), or null
    long long ss;               // floats between consecutive splits
    int splits;                 // 1 .. 127
    __device__ __forceinline__ float operator()(int off) const {
        if constexpr (NS > 0) {         // every partial requested before the first is used: ONE round trip
            float v[NS];
#pragma unroll
            for (int g = 0; g < NS; ++g) v[g] = p[(long long)g * ss + off];
            const float b = bias ? bias[off] : 0.f;
            float tot = 0.f + v[0];
#pragma unroll
            for (int g = 1; g < NS; ++g) tot += 0.f + v[g];
            return bias ? tot + b : tot;
        } else {
            float tot = 0.f;
            for (int g = 0; g < 16 && g < splits; ++g) {
                float sg = 0.f;
                for (int z = g; z < splits; z += 16) sg += p[(long long)z * ss + off];
                tot = g == 0 ? sg : tot + sg;
            }
            return bias ? tot + bias[off] : tot;
        }
    }
};

// Dueling heads (policies/dqn/layers/dueling_merge_layer.py:32-35): the block holds n_actions advantage rows
// followed by ONE value row; logit(a, i) = val_i + (adv_ai - mean_a adv_ai).  Per lane (= atom): the mean and
// the value, read once per sample.
struct Duel { float mean, val; bool on; };
template <class L>
__device__ __forceinline__ Duel duel_terms(const L& logits, int lane, int n_actions, int n_atoms, int stride,
                                           bool dueling) {
    Duel d = {0.f, 0.f, dueling};
    if (dueling && lane < n_atoms) {
        float sum = 0.f;
        for (int a = 0; a < n_actions; ++a) sum += logits(a * stride + lane);
        d.mean = sum / (float)n_actions;
        d.val = logits(n_actions * stride + lane);
    }
    return d;
}
template <class L>
__device__ __forceinline__ float atom_logit(const L& logits, int a, int lane, int n_atoms, int stride,
                                            const Duel& d) {
    if (lane >= n_atoms) return 0.f;
    const float x = logits(a * stride + lane);
    return d.on ? d.val + (x - d.mean) : x;
}

// greedy action of one sample under `logits`: argmax_a sum_i softmax(logit(a, .))_i z_i, first maximum
__device__ __forceinline__ int greedy_action(const float* logits_p, int lane, int n_actions, int n_atoms,
                                             int stride, float z_lane, bool dueling) {
    const Plain logits = {logits_p};
    const Duel d = duel_terms(logits, lane, n_actions, n_atoms, stride, dueling);
    int best = 0;
    float best_q = -3.0e38f;
    for (int a = 0; a < n_actions; ++a) {
        const float q = wave_sum(atom_softmax(atom_logit(logits, a, lane, n_atoms, stride, d), lane, n_atoms) * z_lane);
        if (q > best_q) { best_q = q; best = a; }
    }
    return best;
}

__global__ __launch_bounds__(256) void catdqn_act_kernel(const float* __restrict__ logits,
                                                         const float* __restrict__ z,
                                                         const int32_t* __restrict__ override_or_null,
                                                         int64_t batch, int n_actions, int n_atoms, int stride,
                                                         int dueling, float* __restrict__ onehot,
                                                         uint8_t* __restrict__ greedy) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= batch) return;
    const float z_lane = lane < n_atoms ? z[lane] : 0.f;
    const int g = greedy_action(logits + b * (n_actions + dueling) * stride, lane, n_actions, n_atoms, stride, z_lane,
                                dueling != 0);
    int act = g;
    if (override_or_null && override_or_null[b] >= 0) act = override_or_null[b];
    if (lane < n_actions) onehot[b * n_actions + lane] = lane == act ? 1.f : 0.f;
    if (lane == 0 && greedy) greedy[b] = (uint8_t)g;
}

struct LogitSrc { const float* p; const float* bias; long long ss; int splits; };     // splits == 0: finished logits
struct CatLossArgs {
    const float* pred_logits;       // policy net on obs              [B][A][S]
    const float* tgt_next_logits;   // target net on next_obs         [B][A][S]
    const float* pol_next_logits;   // policy net on next_obs (double DQN) or null
    LogitSrc src[3];                // PARTS: the same three (pred, tgt_next, pol_next), as split partial sums
    const float* z;                 // [n_atoms] support
    const uint8_t* actions;         // [B]
    const float* returns;           // [B] n-step discounted return
    const uint8_t* terminals;       // [B]
    const float* is_weights;        // [B] or null
    float* dlogits;                 // [B][A][S]
    float* loss_rows;               // [B] per-sample (weighted) loss / B
    float* kl;                      // [B] priorities
    int64_t batch;
    int n_actions, n_atoms, stride;
    int dueling;                    // 1: every [A][S] block above is [A + 1][S], the value row last
    float v_min, v_max, gamma_n;    // gamma_n = discount ** reward_horizon (rounded to f32 on the host)
};

// One workgroup per sample.  The greedy next action needs a softmax expectation per action -- a serial chain
// of wave reductions -- so the actions are dealt to the four waves; wave 0 then carries the sample through
// projection, loss and gradient (lane i = atom i).
// wt (PARTS): the workgroups behind the samples' own (blockIdx.x >= batch) write the data gradients' k-contiguous weight
// copies (dgrad_wt_dev.h) for the backward pass that follows -- as the policy-gradient head's launch does (learner.hip)
template <int NS>                  // -1: finished logits; >= 0: split partial sums (Parts<NS>)
__global__ __launch_bounds__(256) void catdqn_loss_kernel(const CatLossArgs a, const arlw::DgradWtArgs wt) {
    constexpr bool PARTS = NS >= 0;
    if (PARTS && (int64_t)blockIdx.x >= a.batch) {
        arlw::dgrad_wt_block(wt, (int)(blockIdx.x - a.batch), (int)threadIdx.x);
        return;
    }
    __shared__ float s_next[64], s_znext[64], s_q[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t b = blockIdx.x;
    const int A = a.n_actions, n = a.n_atoms, S = a.stride;
    const bool duel = a.dueling != 0;
    const int64_t R = (int64_t)(A + a.dueling) * S;      // floats per sample
    const float z_lane = lane < n ? a.z[lane] : 0.f;
    using L = typename std::conditional<PARTS, Parts<(NS > 0 ? NS : 0)>, Plain>::type;
    auto block_of = [&](int which, const float* plain) -> L {
        if constexpr (PARTS) {
            const LogitSrc& q = a.src[which];
            return L{q.p + b * R, q.bias, q.ss, q.splits};
        } else {
            return Plain{plain + b * R};
        }
    };
    const L tgt = block_of(1, a.tgt_next_logits);
    // greedy next action: under the policy net (double DQN) or the target net (cat_dqn.py:77-81), first maximum
    {
        const bool dbl = PARTS ? a.src[2].p != nullptr : a.pol_next_logits != nullptr;
        const L sel = dbl ? block_of(2, a.pol_next_logits) : tgt;
        const Duel d = duel_terms(sel, lane, A, n, S, duel);
        // this wave's actions (k = wave, wave + 4, ...; at most 16 of the <= 64): every logit requested before the first
        // softmax -- a lone global round trip instead of one per action
        float xs[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) xs[j] = wave + 4 * j < A ? atom_logit(sel, wave + 4 * j, lane, n, S, d) : 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = wave + 4 * j;
            if (k < A) {
                const float q = wave_sum(atom_softmax(xs[j], lane, n) * z_lane);
                if (lane == 0) s_q[k] = q;
            }
        }
    }
    // (wave 0's prediction logits do not depend on anything above: on their way across the barrier)
    const int act = a.actions[b];
    const L prd = block_of(0, a.pred_logits);
    float pred_x = 0.f;
    if (wave == 0) pred_x = atom_logit(prd, act, lane, n, S, duel_terms(prd, lane, A, n, S, duel));
    __syncthreads();
    if (wave != 0) return;
    int a_next = 0;
    float best_q = s_q[0];
    for (int k = 1; k < A; ++k) {
        const float q = s_q[k];
        if (q > best_q) { best_q = q; a_next = k; }
    }
    const float next_p = atom_softmax(atom_logit(tgt, a_next, lane, n, S, duel_terms(tgt, lane, A, n, S, duel)), lane, n);
    // shifted support, clipped to [v_min, v_max] (:56-62)
    const float keep = a.terminals[b] ? 0.f : 1.f;
    float zn = a.returns[b] + keep * (a.gamma_n * z_lane);
    zn = fminf(fmaxf(zn, a.v_min), a.v_max);
    s_next[lane] = next_p;              // only wave 0 is left: in-order LDS access of one wave needs no barrier
    s_znext[lane] = zn;
    __builtin_amdgcn_wave_barrier();
    // projection on the base support (:63-70,88-90): proj_i = sum_j next_p_j clip(1 - |zn_j - z_i| / dz, 0, 1)
    const float dz = (a.v_max - a.v_min) / (float)(n - 1);
    float proj = 0.f;
    if (lane < n)
        for (int j = 0; j < n; ++j) {
            const float c = 1.f - fabsf(s_znext[j] - z_lane) / dz;
            proj += s_next[j] * fminf(fmaxf(c, 0.f), 1.f);
        }
    // prediction, cross-entropy with NaN guard (:92-94)
    const float pred = atom_softmax(pred_x, lane, n);
    const float pc = fminf(fmaxf(pred, 1e-6f), 1.f);
    const float w = (a.is_weights ? a.is_weights[b] : 1.f) / (float)a.batch;
    const float ce = lane < n ? -(proj * logf(pc)) : 0.f;
    const float loss_b = wave_sum(ce);
    // KL priority (:100-105)
    const float pj = fminf(fmaxf(proj, 1e-6f), 1.f);
    const float kl_b = wave_sum(lane < n ? pj * logf(pj / pc) : 0.f);
    // gradient: d loss / d pred_i = -w proj_i / pred_i inside the clip range, then softmax backward
    const float g = (lane < n && pred >= 1e-6f && pred <= 1.f) ? -w * proj / pred : 0.f;
    const float dot = wave_sum(g * pred);
    float* dl = a.dlogits + b * R;
    const float d_lane = lane < n ? pred * (g - dot) : 0.f;            // d loss / d logit(act, lane)
    if (!duel) {
        for (int q = lane; q < A * S; q += 64) {                       // every other action (and the padding) gets 0
            const int qa = q / S, qi = q - qa * S;
            if (qa != act || qi >= n) dl[q] = 0.f;
        }
        if (lane < n) dl[act * S + lane] = d_lane;
    } else {                                                           // through the merge: adv rows, then the value row
        const float share = d_lane / (float)A;
        for (int k = 0; k < A; ++k)
            for (int i = lane; i < S; i += 64) dl[k * S + i] = i < n ? (k == act ? d_lane - share : -share) : 0.f;
        for (int i = lane; i < S; i += 64) dl[A * S + i] = i < n ? d_lane : 0.f;
    }
    if (lane == 0) {
        a.loss_rows[b] = w * loss_b;
        a.kl[b] = fminf(fmaxf(kl_b, 1e-6f), 1e6f);
    }
}

// ---- plain DQN: Q rows f32[batch][q_stride], the first n_actions columns valid; one lane per sample ----
// dueling: the row holds n advantages followed by the value; q_a = val + (adv_a - mean adv)
__device__ __forceinline__ float row_mean(const float* row, int n) {
    float sum = 0.f;
    for (int a = 0; a < n; ++a) sum += row[a];
    return sum / (float)n;
}
__device__ __forceinline__ float q_at(const float* row, int a, int n, bool dueling, float mean) {
    return dueling ? row[n] + (row[a] - mean) : row[a];
}
__device__ __forceinline__ int first_argmax(const float* row, int n, bool dueling) {
    const float mean = dueling ? row_mean(row, n) : 0.f;
    int best = 0;
    float best_q = q_at(row, 0, n, dueling, mean);
    for (int a = 1; a < n; ++a) {
        const float q = q_at(row, a, n, dueling, mean);
        if (q > best_q) { best_q = q; best = a; }
    }
    return best;
}

__global__ __launch_bounds__(256) void dqn_act_kernel(const float* __restrict__ q,
                                                      const int32_t* __restrict__ override_or_null, int64_t batch,
                                                      int n_actions, int stride, int dueling,
                                                      float* __restrict__ onehot, uint8_t* __restrict__ greedy) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const int g = first_argmax(q + b * stride, n_actions, dueling != 0);
    int act = g;
    if (override_or_null && override_or_null[b] >= 0) act = override_or_null[b];
    for (int a = 0; a < n_actions; ++a) onehot[b * n_actions + a] = a == act ? 1.f : 0.f;
    if (greedy) greedy[b] = (uint8_t)g;
}

struct DqnLossArgs {
    const float* q;                 // policy net on obs              [B][S]
    const float* tgt_next_q;        // target net on next_obs         [B][S]
    const float* pol_next_q;        // policy net on next_obs (double DQN) or null
    const uint8_t* actions;         // [B]
    const float* returns;           // [B] n-step discounted return
    const uint8_t* terminals;       // [B]
    const float* is_weights;        // [B] or null
    float* dq;                      // [B][S]
    float* loss_rows;               // [B] per-sample (weighted) loss / B
    float* td_abs;                  // [B] priorities: |TD error| clipped to delta_clip
    int64_t batch;
    int n_actions, stride;
    int dueling;                    // 1: column n_actions of every row is the value stream
    float gamma_n, delta_clip;      // delta_clip <= 0: squared loss, unclipped priorities
};

__global__ __launch_bounds__(256) void dqn_loss_kernel(const DqnLossArgs a) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.batch) return;
    const int A = a.n_actions, S = a.stride;
    const float* tgt = a.tgt_next_q + b * S;
    // dqn.py:146-150: double DQN picks the action with the policy net and values it with the target net
    const bool duel = a.dueling != 0;
    const int a_next = first_argmax(a.pol_next_q ? a.pol_next_q + b * S : tgt, A, duel);
    const float next_q = q_at(tgt, a_next, A, duel, duel ? row_mean(tgt, A) : 0.f);
    const float keep = a.terminals[b] ? 0.f : 1.f;
    const float y = a.returns[b] + keep * (a.gamma_n * next_q);            // :152-153
    const int act = a.actions[b];
    const float* qrow = a.q + b * S;
    const float d = y - q_at(qrow, act, A, duel, duel ? row_mean(qrow, A) : 0.f);
    const float ad = fabsf(d), c = a.delta_clip;
    float loss = 0.5f * (d * d), slope = d;                                // d loss / d d
    if (c > 0.f && ad > c) {                                               // Huber (:157-160)
        loss = c * (ad - c / 2.f);
        slope = d > 0.f ? c : -c;
    }
    const float w = (a.is_weights ? a.is_weights[b] : 1.f) / (float)a.batch;
    float* dq = a.dq + b * S;
    const float gq = -(w * slope);                                         // d = y - q, y carries no gradient
    for (int k = 0; k < S; ++k) dq[k] = 0.f;
    if (!duel) {
        dq[act] = gq;
    } else {                                                               // through the merge
        const float share = gq / (float)A;
        for (int k = 0; k < A; ++k) dq[k] = k == act ? gq - share : -share;
        dq[A] = gq;
    }
    a.loss_rows[b] = w * loss;
    a.td_abs[b] = c > 0.f ? fminf(ad, c) : ad;                             // :165
}

}  // namespace

static int check_q(int64_t batch, int n_actions, int stride, int dueling) {
    if (batch <= 0 || n_actions <= 0 || n_actions > 255 || stride < n_actions + (dueling != 0) || (stride & 3)) {
        arl::set_error("dqn: need batch > 0, 1 <= n_actions <= 255, q_stride >= n_actions (+ 1 if dueling) and %% 4 == 0");
        return ARL_E_RANGE;
    }
    return 0;
}

extern "C" int arl_dqn_act(const float* q, const int32_t* override_or_null, int64_t batch, int32_t n_actions,
                           int32_t q_stride, int32_t dueling, float* onehot, uint8_t* greedy_or_null, void* stream) {
    ARL_REQUIRE(q && onehot, ARL_E_ARG, "null pointer");
    int rc = check_q(batch, n_actions, q_stride, dueling);
    if (rc) return rc;
    hipLaunchKernelGGL(dqn_act_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       q, override_or_null, batch, n_actions, q_stride, dueling != 0, onehot, greedy_or_null);
    return arl::check_launch("dqn_act_kernel");
}

extern "C" int arl_dqn_loss(const float* q, const float* tgt_next_q, const float* pol_next_q_or_null,
                            const uint8_t* actions, const float* returns, const uint8_t* terminals,
                            const float* is_weights_or_null, int64_t batch, int32_t n_actions, int32_t q_stride,
                            int32_t dueling, float gamma_n, float delta_clip, float* dq, float* loss_rows,
                            float* td_abs, void* stream) {
    ARL_REQUIRE(q && tgt_next_q && actions && returns && terminals && dq && loss_rows && td_abs, ARL_E_ARG,
                "null pointer");
    int rc = check_q(batch, n_actions, q_stride, dueling);
    if (rc) return rc;
    DqnLossArgs a = {};
    a.q = q; a.tgt_next_q = tgt_next_q; a.pol_next_q = pol_next_q_or_null; a.actions = actions; a.returns = returns;
    a.terminals = terminals; a.is_weights = is_weights_or_null; a.dq = dq; a.loss_rows = loss_rows; a.td_abs = td_abs;
    a.batch = batch; a.n_actions = n_actions; a.stride = q_stride; a.dueling = dueling != 0; a.gamma_n = gamma_n;
    a.delta_clip = delta_clip;
    hipLaunchKernelGGL(dqn_loss_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return arl::check_launch("dqn_loss_kernel");
}

static int check_cat(int64_t batch, int n_actions, int n_atoms, int stride) {
    if (batch <= 0 || n_actions <= 0 || n_actions > 64 || n_atoms < 2 || n_atoms > 64 || stride < n_atoms || (stride & 3)) {
        arl::set_error("catdqn: need batch > 0, 1 <= n_actions <= 64, 2 <= n_atoms <= 64, atom_stride >= n_atoms and %% 4 == 0");
        return ARL_E_RANGE;
    }
    return 0;
}

extern "C" int arl_catdqn_act(const float* logits, const float* z, const int32_t* override_or_null, int64_t batch,
                              int32_t n_actions, int32_t n_atoms, int32_t atom_stride, int32_t dueling, float* onehot,
                              uint8_t* greedy_or_null, void* stream) {
    ARL_REQUIRE(logits && z && onehot, ARL_E_ARG, "null pointer");
    int rc = check_cat(batch, n_actions, n_atoms, atom_stride);
    if (rc) return rc;
    hipLaunchKernelGGL(catdqn_act_kernel, dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       logits, z, override_or_null, batch, n_actions, n_atoms, atom_stride, dueling != 0, onehot,
                       greedy_or_null);
    return arl::check_launch("catdqn_act_kernel");
}

extern "C" int arl_catdqn_loss(const float* pred_logits, const float* tgt_next_logits, const float* pol_next_logits_or_null,
                               const float* z, const uint8_t* actions, const float* returns, const uint8_t* terminals,
                               const float* is_weights_or_null, int64_t batch, int32_t n_actions, int32_t n_atoms,
                               int32_t atom_stride, int32_t dueling, float v_min, float v_max, float gamma_n,
                               float* dlogits, float* loss_rows, float* kl, void* stream) {
    ARL_REQUIRE(pred_logits && tgt_next_logits && z && actions && returns && terminals && dlogits && loss_rows && kl,
                ARL_E_ARG, "null pointer");
    int rc = check_cat(batch, n_actions, n_atoms, atom_stride);
    if (rc) return rc;
    ARL_REQUIRE(v_max > v_min, ARL_E_ARG, "v_max must exceed v_min");
    CatLossArgs a = {};
    a.pred_logits = pred_logits; a.tgt_next_logits = tgt_next_logits; a.pol_next_logits = pol_next_logits_or_null;
    a.z = z; a.actions = actions; a.returns = returns; a.terminals = terminals; a.is_weights = is_weights_or_null;
    a.dlogits = dlogits; a.loss_rows = loss_rows; a.kl = kl; a.batch = batch;
    a.n_actions = n_actions; a.n_atoms = n_atoms; a.stride = atom_stride; a.dueling = dueling != 0;
    a.v_min = v_min; a.v_max = v_max; a.gamma_n = gamma_n;
    hipLaunchKernelGGL(catdqn_loss_kernel<-1>, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, a, arlw::DgradWtArgs{});
    return arl::check_launch("catdqn_loss_kernel");
}

extern "C" int arl_catdqn_loss_parts(const arl_logit_src* pred, const arl_logit_src* tgt_next,
                                     const arl_logit_src* pol_next_or_null, const float* z, const uint8_t* actions,
                                     const float* returns, const uint8_t* terminals, const float* is_weights_or_null,
                                     int64_t batch, int32_t n_actions, int32_t n_atoms, int32_t atom_stride,
                                     int32_t dueling, float v_min, float v_max, float gamma_n, float* dlogits,
                                     float* loss_rows, float* kl, const arl_dgrad_wt* wt_items_or_null, int32_t n_wt,
                                     void* stream) {
    ARL_REQUIRE(pred && tgt_next && pred->part && tgt_next->part && z && actions && returns && terminals && dlogits &&
                    loss_rows && kl && (!pol_next_or_null || pol_next_or_null->part), ARL_E_ARG, "null pointer");
    int rc = check_cat(batch, n_actions, n_atoms, atom_stride);
    if (rc) return rc;
    ARL_REQUIRE(v_max > v_min, ARL_E_ARG, "v_max must exceed v_min");
    const arl_logit_src* in[3] = {pred, tgt_next, pol_next_or_null};
    CatLossArgs a = {};
    for (int i = 0; i < 3; ++i) {
        if (!in[i]) continue;
        ARL_REQUIRE(in[i]->splits >= 1 && in[i]->splits < 128 && in[i]->split_stride > 0, ARL_E_RANGE,
                    "logit source: 1 <= splits < 128 partial sums, split_stride > 0 (finished logits: one split, no bias)");
        a.src[i].p = in[i]->part; a.src[i].bias = in[i]->bias_or_null; a.src[i].ss = in[i]->split_stride;
        a.src[i].splits = in[i]->splits;
    }
    a.z = z; a.actions = actions; a.returns = returns; a.terminals = terminals; a.is_weights = is_weights_or_null;
    a.dlogits = dlogits; a.loss_rows = loss_rows; a.kl = kl; a.batch = batch;
    a.n_actions = n_actions; a.n_atoms = n_atoms; a.stride = atom_stride; a.dueling = dueling != 0;
    a.v_min = v_min; a.v_max = v_max; a.gamma_n = gamma_n;
    // all sources on the same power-of-two split count (the usual case: the same layer at two batch sizes): unrolled loads
    int ns = a.src[0].splits;
    for (int i = 1; i < 3; ++i)
        if (in[i] && a.src[i].splits != ns) ns = 0;
    arlw::DgradWtArgs wt = {};
    int wt_blocks = 0;
    if (wt_items_or_null && n_wt > 0) {
        rc = arlw::dgrad_wt_plan(wt_items_or_null, n_wt, &wt, &wt_blocks);
        if (rc) return rc;
    }
    const dim3 grid((unsigned)(batch + wt_blocks));
    hipStream_t st = (hipStream_t)stream;
    switch (ns) {
    case 1: hipLaunchKernelGGL(catdqn_loss_kernel<1>, grid, dim3(256), 0, st, a, wt); break;
    case 2: hipLaunchKernelGGL(catdqn_loss_kernel<2>, grid, dim3(256), 0, st, a, wt); break;
    case 4: hipLaunchKernelGGL(catdqn_loss_kernel<4>, grid, dim3(256), 0, st, a, wt); break;
    case 8: hipLaunchKernelGGL(catdqn_loss_kernel<8>, grid, dim3(256), 0, st, a, wt); break;
    case 16: hipLaunchKernelGGL(catdqn_loss_kernel<16>, grid, dim3(256), 0, st, a, wt); break;
    default: hipLaunchKernelGGL(catdqn_loss_kernel<0>, grid, dim3(256), 0, st, a, wt);
    }
    return arl::check_launch("catdqn_loss_kernel (parts)");
}
