// Device replay memory for the DQN family: frame-dedup ring buffers for all
// environments in HBM, n-step return back-fill, batch extraction, and the f64 parted sum tree of
// prioritized replay.  HBM-bound byte / index work, no MFMA.
//
//   arl_replay_append     FrameReplayBuffer.append_data / EnvBuffer.write_samples
//                         (accel_rl/algos/dqn/replay_buffers/frame.py:57-60,121-166)
//   arl_replay_extract    extract_batch / extract_observations            (frame.py:69-90)
//   arl_sumtree_find      PartedSumTree.find                              (sum_tree.py:88-98)
//   arl_sumtree_add       PartedSumTree.reconstruct (np.add.at, in input order, :54-57)
//   arl_sumtree_gather    tree[idxs]                                      (:83,:65)
//
// Layout (struct arl_replay): per environment a ring of `size` states; frames are stored ONCE
// each, u8[n_env][size + F - 1][frame_bytes] (the last F-1 slots mirror the first after a wrap),
// so a stacked observation is F consecutive slots = one contiguous F*frame_bytes run.

#include "arl_common.h"

namespace {

struct AppendArgs {
    arl_replay rb;
    const uint8_t* observations;   // [n_env*T][F][frame_bytes]
    const uint8_t* actions;        // [n_env*T]
    const float* rewards;
    const uint8_t* dones;
    int horizon, idx, promo;
    double disc_pow[ARL_REPLAY_MAX_HORIZON];    // discount^i (NEP50: already rounded to f32)
};

__device__ __forceinline__ void copy_bytes16(uint8_t* dst, const uint8_t* src, int bytes, int tid, int nthreads) {
    const uint4* s = reinterpret_cast<const uint4*>(src);
    uint4* d = reinterpret_cast<uint4*>(dst);
    for (int i = tid; i < bytes / 16; i += nthreads) d[i] = s[i];
}

// one workgroup per environment
__global__ __launch_bounds__(256) void replay_append_kernel(const AppendArgs a) {
    const int e = blockIdx.x, tid = threadIdx.x;
    const int S = a.rb.size, F = a.rb.n_stack, P = a.rb.frame_bytes, T = a.horizon, idx = a.idx, h_r = a.rb.reward_horizon;
    const int ring = S + F - 1;
    uint8_t* frames = a.rb.frames + (int64_t)e * ring * P;
    uint8_t* nb = a.rb.n_blanks + (int64_t)e * ring;
    uint8_t* acts = a.rb.acts + (int64_t)e * S;
    uint8_t* term = a.rb.terminals + (int64_t)e * S;
    float* rew = a.rb.rewards + (int64_t)e * S;
    float* ret = a.rb.returns + (int64_t)e * S;
    if (idx == 0) {                                   // mirror the ring's tail (frame.py:134-137)
        copy_bytes16(frames, frames + (int64_t)S * P, (F - 1) * P, tid, 256);
        if (tid < F - 1) nb[tid] = nb[S + tid];
    }
    for (int t = 0; t < T; ++t)                       // newest frame of every step (:142-143)
        copy_bytes16(frames + (int64_t)(idx + F - 1 + t) * P,
                     a.observations + ((int64_t)(e * T + t) * F + (F - 1)) * P, P, tid, 256);
    if (tid < T) {
        acts[idx + tid] = a.actions[e * T + tid];
        rew[idx + tid] = a.rewards[e * T + tid];
        term[idx + tid] = a.dones[e * T + tid] ? 1 : 0;
    }
    __syncthreads();
    if (tid == 0) {                                   // blank-history marks (:144-155), in step order
        for (int t = 0; t < T; ++t) {
            const int p = idx + t;
            if (a.dones[e * T + t]) {                 // the sampler's dones, not the propagated terminals
                for (int k = 1; k < F; ++k) nb[p + k] = (uint8_t)(F - k);
            } else if (nb[p + 1] && nb[p + 1] >= nb[p]) {
                nb[p + 1] = 0;
            }
        }
    }
    if (tid == 64) {                                  // n-step returns h_r - 1 behind (:156-166), in step order
        for (int t = 0; t < T; ++t) {
            int j = idx - (h_r - 1) + t;
            if (j < 0) j += S;
            float r32 = rew[j];
            double r64 = (double)rew[j];
            if (!term[j]) {
                for (int i = 1; i < h_r; ++i) {
                    int k = j + i;
                    if (k >= S) k -= S;
                    if (a.promo == ARL_PROMO_NEP50) r32 = r32 + (float)a.disc_pow[i] * rew[k];
                    else r64 = r64 + a.disc_pow[i] * (double)rew[k];
                    if (term[k]) { term[j] = 1; break; }
                }
            }
            ret[j] = a.promo == ARL_PROMO_NEP50 ? r32 : (float)r64;
        }
    }
}

struct ExtractArgs {
    arl_replay rb;
    const int32_t* env_idxs;
    const int32_t* step_idxs;
    uint8_t* obs;          // [batch][F][frame_bytes]
    uint8_t* next_obs;
    uint8_t* actions;
    float* returns;
    uint8_t* terminals;
    int64_t batch;
};

// grid (batch, 2): one workgroup copies one stacked observation (F * frame_bytes contiguous)
// and zeroes its leading n_blanks frames (frame.py:81-90)
__global__ __launch_bounds__(256) void replay_extract_kernel(const ExtractArgs a) {
    const int64_t j = blockIdx.x;
    const int which = blockIdx.y, tid = threadIdx.x;
    const int S = a.rb.size, F = a.rb.n_stack, P = a.rb.frame_bytes, ring = S + F - 1;
    const int e = a.env_idxs[j];
    int i = a.step_idxs[j];
    if (which) { i += a.rb.reward_horizon; if (i >= S) i -= S; }     // frame.py:70
    const uint8_t* src = a.rb.frames + ((int64_t)e * ring + i) * P;
    uint8_t* dst = (which ? a.next_obs : a.obs) + j * F * P;
    const int blanks = a.rb.n_blanks[(int64_t)e * ring + i];
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    const int per_frame = P / 16, total = F * per_frame, zero_upto = blanks * per_frame;
    for (int q = tid; q < total; q += 256) d4[q] = q < zero_upto ? make_uint4(0, 0, 0, 0) : s4[q];
    if (!which && tid == 0) {
        a.actions[j] = a.rb.acts[(int64_t)e * S + a.step_idxs[j]];
        a.returns[j] = a.rb.returns[(int64_t)e * S + a.step_idxs[j]];
        a.terminals[j] = a.rb.terminals[(int64_t)e * S + a.step_idxs[j]];
    }
}

// ------------------------------------------------------------------------------------------ sum tree
__global__ __launch_bounds__(256) void sumtree_find_kernel(const double* __restrict__ tree, int levels,
                                                           const double* __restrict__ uniforms, int64_t n,
                                                           int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = uniforms[i] * tree[0];
    int idx = 0;
    for (int l = 0; l < levels - 1; ++l) {
        idx = 2 * idx + 1;
        const double left = tree[idx];
        if (v > left) { v -= left; idx += 1; }
    }
    out[i] = idx;
}

__global__ __launch_bounds__(256) void sumtree_gather_kernel(const double* __restrict__ tree,
                                                             const int32_t* __restrict__ idxs, int64_t n,
                                                             double scale, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = scale * tree[idxs[i]];
}

// np.add.at(tree, idxs >> level, diffs) for every level, several updates of one node applied in INPUT
// order (that fixes the f64 rounding, sum_tree.py:54-57).  The levels touch disjoint nodes, so each gets
// its own workgroup (n <= 4096 items in LDS): the first item that names a node adds that node's whole run,
// scanning the later items in order.  O(n^2 / 1024) LDS compares per thread, no sort, no atomics.
constexpr int ADD_MAX = 4096;

// priorities != null (arl_sumtree_update_pow): diffs[i] = priority_diffs_kernel's value, computed here
__global__ __launch_bounds__(1024) void sumtree_add_kernel(double* __restrict__ tree,
                                                           const int32_t* __restrict__ idxs,
                                                           const double* __restrict__ diffs, int n,
                                                           const float* __restrict__ priorities = nullptr,
                                                           const double* __restrict__ last_probs = nullptr,
                                                           double alpha = 0.0) {
    __shared__ __attribute__((aligned(16))) unsigned s_node[ADD_MAX];
    __shared__ double s_diff[ADD_MAX];
    const int tid = threadIdx.x, l = blockIdx.x;                 // one workgroup per tree level
    for (int i = tid; i < n; i += 1024) {
        s_node[i] = (unsigned)((idxs[i] + 1) >> l) - 1u;         // parent = (i - 1) / 2, l times
        s_diff[i] = priorities ? (double)(float)pow((double)priorities[i], (double)(float)alpha) - last_probs[i] : diffs[i];
    }
    __syncthreads();
    const uint4* node4 = reinterpret_cast<const uint4*>(s_node);
    for (int i = tid; i < n; i += 1024) {
        const unsigned node = s_node[i];
        bool first = true;
        const int i4 = i >> 2;
        for (int q = 0; q < i4 && first; ++q) {                  // four earlier items per LDS read
            const uint4 v = node4[q];
            first = !(v.x == node || v.y == node || v.z == node || v.w == node);
        }
        for (int j = i4 << 2; j < i && first; ++j) first = s_node[j] != node;
        if (first) {                                             // this item opens its node's run: add the run in input order
            double t = tree[node] + s_diff[i];
            int j = i + 1;
            for (; (j & 3) && j < n; ++j)
                if (s_node[j] == node) t += s_diff[j];
            for (; j + 3 < n; j += 4) {
                const uint4 v = node4[j >> 2];
                if (v.x == node) t += s_diff[j];
                if (v.y == node) t += s_diff[j + 1];
                if (v.z == node) t += s_diff[j + 2];
                if (v.w == node) t += s_diff[j + 3];
            }
            for (; j < n; ++j)
                if (s_node[j] == node) t += s_diff[j];
            tree[node] = t;
        }
    }
}

// sample_n's device half (sum_tree.py:77-86): descend for every uniform, sort, drop duplicates, keep the n
// smallest distinct leaves with their probabilities and (part, step) coordinates, and report how many distinct
// leaves there were -- the host only needs that count to decide whether the reference would have drawn more.
// One workgroup; the candidates (m <= 4096) are sorted in LDS by a bitonic network.
constexpr int SAMPLE_MAX = 4096;

__global__ __launch_bounds__(1024) void sumtree_sample_kernel(const double* __restrict__ tree, int levels,
                                                              const double* __restrict__ uniforms, int m, int n,
                                                              int part_size, int32_t* __restrict__ tree_idxs,
                                                              int32_t* __restrict__ env_idxs,
                                                              int32_t* __restrict__ step_idxs,
                                                              double* __restrict__ probs,
                                                              int32_t* __restrict__ n_unique, double beta,
                                                              float* __restrict__ is_weights,
                                                              volatile long long* notify, long long ticket) {
    __shared__ int s_key[SAMPLE_MAX];
    __shared__ int s_pos[SAMPLE_MAX];
    __shared__ double s_max[1024];
    // The descent is a chain of dependent loads, one per level (21 for a million leaves, ~0.5 us each from L2): the top
    // TOP_LEVELS of the tree (2^TOP_LEVELS - 1 nodes, one coalesced pass) wait in LDS, where a level costs a few cycles.
    constexpr int TOP_LEVELS = 11, TOP_NODES = (1 << TOP_LEVELS) - 1;
    __shared__ double s_top[TOP_NODES];
    const int tid = threadIdx.x, NT = blockDim.x;         // NT: a power of two, P / NT <= 4
    int P = 1;
    while (P < m) P <<= 1;
    const int top_levels = levels < TOP_LEVELS ? levels : TOP_LEVELS;
    // the uniforms may lie in page-locked HOST memory (arl_sumtree_sample_batch: ~2 us away): on their way while the top
    // of the tree is fetched (P / NT <= 4 candidates per thread)
    double u_pre[SAMPLE_MAX / 1024];
    for (int i = tid, q = 0; i < P; i += NT, ++q) u_pre[q] = i < m ? uniforms[i] : 0.0;
    {   // eight loads in flight per thread and round (a lone wave -- the 33-candidate batch -- would otherwise walk the
        // 2 047 nodes as 32 dependent round trips: the compiler does not hoist the loads over the LDS stores)
        const int nodes = (1 << top_levels) - 1;
        for (int base = 0; base < nodes; base += 8 * NT) {
            double t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int i = base + j * NT + tid; t[j] = i < nodes ? tree[i] : 0.0; }
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int i = base + j * NT + tid; if (i < nodes) s_top[i] = t[j]; }
        }
    }
    __syncthreads();
    const double root = s_top[0];
    for (int i = tid, q = 0; i < P; i += NT, ++q) {
        int idx = 0x7fffffff;                                    // padding sorts behind every leaf
        if (i < m) {
            double v = u_pre[q] * root;
            idx = 0;
            int l = 0;
            for (; l < levels - 1 && l + 1 < top_levels; ++l) {  // node idx is on level l + 1: in LDS
                idx = 2 * idx + 1;
                const double left = s_top[idx];
                if (v > left) { v -= left; idx += 1; }
            }
            // below the cached levels a level is a dependent load from L2 / HBM: TWO levels per round trip -- the left
            // child and both candidates for the level under it are fetched together (same comparisons and
            // subtractions in the same order: the same leaf)
            for (; l + 1 < levels - 1; l += 2) {
                const int c = 2 * idx + 1;
                const double left = tree[c], left_l = tree[2 * c + 1], left_r = tree[2 * c + 3];
                double left2;
                if (v > left) { v -= left; idx = c + 1; left2 = left_r; } else { idx = c; left2 = left_l; }
                idx = 2 * idx + 1;
                if (v > left2) { v -= left2; idx += 1; }
            }
            if (l < levels - 1) {
                idx = 2 * idx + 1;
                const double left = tree[idx];
                if (v > left) { v -= left; idx += 1; }
            }
        }
        s_key[i] = idx;
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += NT) {
                const int partner = i ^ j;
                if (partner > i) {
                    const int a = s_key[i], b = s_key[partner];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { s_key[i] = b; s_key[partner] = a; }
                }
            }
            __syncthreads();
        }
    // distinct leaves: flag the first of every run, inclusive scan of the flags (Hillis-Steele in LDS)
    for (int i = tid; i < P; i += NT)
        s_pos[i] = (s_key[i] != 0x7fffffff && (i == 0 || s_key[i] != s_key[i - 1])) ? 1 : 0;
    __syncthreads();
    for (int off = 1; off < P; off <<= 1) {
        int add[SAMPLE_MAX / 1024];
        for (int i = tid, q = 0; i < P; i += NT, ++q) add[q] = i >= off ? s_pos[i - off] : 0;
        __syncthreads();
        for (int i = tid, q = 0; i < P; i += NT, ++q) s_pos[i] += add[q];
        __syncthreads();
    }
    const int total = s_pos[P - 1];
    if (tid == 0) {
        n_unique[0] = total;
        if (notify) {
            // the host waits on this word in page-locked memory instead of a copy + event; all it needs is the count, known
            // here -- the uniforms were consumed before the sort -- so it is told now and reacts (launches the update's
            // graph) while the probabilities and weights below are still being written: every output is queued work's
            // business (stream order).  No system-scope fence: the word carries everything the host reads directly (a
            // fence here writes the XCD's dirty L2 back first -- microseconds the update's critical path waits for)
            *notify = (ticket << 32) | (long long)(unsigned)total;
        }
    }
    const int shift = (1 << (levels - 1)) - 1;                   // tree index of leaf 0
    // importance-sampling weights of the batch (is_weights_kernel's arithmetic: w = (1 / p) ** beta in f64, divided by
    // their maximum, rounded to f32), from the probabilities while they are in registers: wq[q] = this thread's q-th slot
    double wq[SAMPLE_MAX / 1024], wmax = 0.0;
    int wpos[SAMPLE_MAX / 1024];
    for (int i = tid, q = 0; i < P; i += NT, ++q) {
        const int key = s_key[i];
        const bool first = key != 0x7fffffff && (i == 0 || key != s_key[i - 1]);
        const int pos = s_pos[i] - 1;
        wpos[q] = -1;
        if (first && pos < n) {
            const double pr = tree[key];
            tree_idxs[pos] = key;
            probs[pos] = pr;
            const int leaf = key - shift;
            env_idxs[pos] = leaf / part_size;
            step_idxs[pos] = leaf - (leaf / part_size) * part_size;
            if (is_weights) { wq[q] = pow(1.0 / pr, beta); wpos[q] = pos; wmax = fmax(wmax, wq[q]); }
        }
    }
    // too few distinct leaves: the slots past them repeat the first one, so that work already queued behind this
    // kernel (batch extraction) stays in bounds while the host learns from n_unique that it has to top up
    for (int pos = total + tid; pos < n; pos += NT) {
        const int key = s_key[0], leaf = key - shift;
        tree_idxs[pos] = key;
        probs[pos] = tree[key];
        env_idxs[pos] = leaf / part_size;
        step_idxs[pos] = leaf - (leaf / part_size) * part_size;
    }
    if (is_weights) {                                            // (meaningful only with n distinct leaves: the caller checks)
        s_max[tid] = wmax;
        __syncthreads();
        for (int off = NT >> 1; off > 0; off >>= 1) {
            if (tid < off) s_max[tid] = fmax(s_max[tid], s_max[tid + off]);
            __syncthreads();
        }
        const double mx = s_max[0];
        for (int i = tid, q = 0; i < P; i += NT, ++q)
            if (wpos[q] >= 0) is_weights[wpos[q]] = (float)(wq[q] / mx);
        for (int pos = total + tid; pos < n; pos += NT) is_weights[pos] = 0.f;
    }
}

// importance-sampling weights (prioritized.py:33-35): w = (1 / p) ** beta in f64, divided by their maximum,
// rounded to f32 for the loss kernels.  One workgroup.
__global__ __launch_bounds__(1024) void is_weights_kernel(const double* __restrict__ probs, int n, double beta,
                                                          float* __restrict__ out) {
    __shared__ double s_max[1024];
    double mx = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) mx = fmax(mx, pow(1.0 / probs[i], beta));
    s_max[threadIdx.x] = mx;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) s_max[threadIdx.x] = fmax(s_max[threadIdx.x], s_max[threadIdx.x + off]);
        __syncthreads();
    }
    mx = s_max[0];
    for (int i = threadIdx.x; i < n; i += 1024) out[i] = (float)(pow(1.0 / probs[i], beta) / mx);
}

// update_batch_priorities' arithmetic (prioritized.py:37-38, sum_tree.py:74-75): the f32 priorities raised to
// alpha in f32 (numpy keeps float32 ** python-float in float32), minus the probabilities sampled before, in f64
__global__ __launch_bounds__(256) void priority_diffs_kernel(const float* __restrict__ priorities,
                                                             const double* __restrict__ last_probs, int64_t n,
                                                             double alpha, double* __restrict__ diffs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) diffs[i] = (double)(float)pow((double)priorities[i], (double)(float)alpha) - last_probs[i];
}

int check_replay(const arl_replay* rb) {
    if (!rb || !rb->frames || !rb->acts || !rb->n_blanks || !rb->terminals || !rb->rewards || !rb->returns) {
        arl::set_error("replay: null pointer");
        return ARL_E_ARG;
    }
    if (rb->n_env <= 0 || rb->size <= 0 || rb->n_stack < 2 || rb->frame_bytes <= 0 || (rb->frame_bytes & 15) ||
        rb->reward_horizon < 1 || rb->reward_horizon > ARL_REPLAY_MAX_HORIZON || rb->reward_horizon > rb->size) {
        arl::set_error("replay: need n_stack >= 2, frame_bytes %% 16 == 0, 1 <= reward_horizon <= %d", ARL_REPLAY_MAX_HORIZON);
        return ARL_E_RANGE;
    }
    if (!arl::aligned16(rb->frames)) { arl::set_error("replay: frames must be 16-byte aligned"); return ARL_E_ALIGN; }
    return 0;
}

}  // namespace

extern "C" int arl_replay_append(const arl_replay* rb, const uint8_t* observations, const uint8_t* actions,
                                 const float* rewards, const uint8_t* dones, int32_t horizon, int32_t idx,
                                 double discount, int32_t promo, void* stream) {
    int rc = check_replay(rb);
    if (rc) return rc;
    ARL_REQUIRE(observations && actions && rewards && dones, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(horizon > 0 && horizon <= 256 && rb->size % horizon == 0 && idx >= 0 && idx + horizon <= rb->size &&
                    idx % horizon == 0, ARL_E_RANGE, "size must be a multiple of horizon (<= 256), idx a multiple of it");
    ARL_REQUIRE(promo == ARL_PROMO_NEP50 || promo == ARL_PROMO_LEGACY, ARL_E_ARG, "bad promo");
    ARL_REQUIRE(arl::aligned16(observations), ARL_E_ALIGN, "observations must be 16-byte aligned");
    AppendArgs a = {};
    a.rb = *rb; a.observations = observations; a.actions = actions; a.rewards = rewards; a.dones = dones;
    a.horizon = horizon; a.idx = idx; a.promo = promo;
    double p = 1.0;
    for (int i = 0; i < ARL_REPLAY_MAX_HORIZON; ++i) {       // discount ** i, as Python computes it (repeated
        a.disc_pow[i] = promo == ARL_PROMO_NEP50 ? (double)(float)p : p;     // multiplication == pow for i <= 2;
        p = pow(discount, i + 1);                                            // use pow() itself to be exact)
    }
    hipLaunchKernelGGL(replay_append_kernel, dim3((unsigned)rb->n_env), dim3(256), 0, (hipStream_t)stream, a);
    return arl::check_launch("replay_append_kernel");
}

extern "C" int arl_replay_extract(const arl_replay* rb, const int32_t* env_idxs, const int32_t* step_idxs,
                                  int64_t batch, uint8_t* obs, uint8_t* next_obs, uint8_t* actions,
                                  float* returns, uint8_t* terminals, void* stream) {
    int rc = check_replay(rb);
    if (rc) return rc;
    ARL_REQUIRE(env_idxs && step_idxs && obs && next_obs && actions && returns && terminals, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(batch >= 0 && batch < ((int64_t)1 << 31), ARL_E_RANGE, "bad batch");
    ARL_REQUIRE(arl::aligned16(obs) && arl::aligned16(next_obs), ARL_E_ALIGN, "obs buffers must be 16-byte aligned");
    if (batch == 0) return 0;
    ExtractArgs a = {};
    a.rb = *rb; a.env_idxs = env_idxs; a.step_idxs = step_idxs; a.obs = obs; a.next_obs = next_obs;
    a.actions = actions; a.returns = returns; a.terminals = terminals; a.batch = batch;
    hipLaunchKernelGGL(replay_extract_kernel, dim3((unsigned)batch, 2), dim3(256), 0, (hipStream_t)stream, a);
    return arl::check_launch("replay_extract_kernel");
}

extern "C" int arl_sumtree_find(const double* tree, int32_t levels, const double* uniforms, int64_t n,
                                int32_t* tree_idxs, void* stream) {
    ARL_REQUIRE(tree && uniforms && tree_idxs, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(levels >= 1 && levels <= 31 && n >= 0, ARL_E_RANGE, "bad levels / n");
    if (n == 0) return 0;
    hipLaunchKernelGGL(sumtree_find_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       tree, levels, uniforms, n, tree_idxs);
    return arl::check_launch("sumtree_find_kernel");
}

// threads of sumtree_sample_kernel: one per padded candidate, 64 .. 1024 (a 33-candidate batch is ONE wave: its ~40
// barriers cost a wave nothing and sixteen waves ~8 us)
static int sample_threads(int m) {
    int p = 64;
    while (p < m && p < 1024) p <<= 1;
    return p;
}

extern "C" int arl_sumtree_sample(const double* tree, int32_t levels, const double* uniforms, int32_t m, int32_t n,
                                  int32_t part_size, int32_t* tree_idxs, int32_t* env_idxs, int32_t* step_idxs,
                                  double* probs, int32_t* n_unique, void* stream) {
    ARL_REQUIRE(tree && uniforms && tree_idxs && env_idxs && step_idxs && probs && n_unique, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(levels >= 1 && levels <= 31 && m >= 1 && m <= SAMPLE_MAX && n >= 1 && n <= m && part_size >= 1,
                ARL_E_RANGE, "need 1 <= n <= m <= 4096 candidates");
    hipLaunchKernelGGL(sumtree_sample_kernel, dim3(1), dim3(sample_threads(m)), 0, (hipStream_t)stream, tree, levels, uniforms,
                       m, n, part_size, tree_idxs, env_idxs, step_idxs, probs, n_unique, 0.0, (float*)nullptr,
                       (volatile long long*)nullptr, 0ll);
    return arl::check_launch("sumtree_sample_kernel");
}

extern "C" int arl_sumtree_sample_batch(const double* tree, int32_t levels, const double* uniforms, int32_t m, int32_t n,
                                        int32_t part_size, int32_t* tree_idxs, int32_t* env_idxs, int32_t* step_idxs,
                                        double* probs, int32_t* n_unique, double beta, float* is_weights_or_null,
                                        int64_t* notify_or_null, int32_t ticket, void* stream) {
    ARL_REQUIRE(tree && uniforms && tree_idxs && env_idxs && step_idxs && probs && n_unique, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(levels >= 1 && levels <= 31 && m >= 1 && m <= SAMPLE_MAX && n >= 1 && n <= m && part_size >= 1,
                ARL_E_RANGE, "need 1 <= n <= m <= 4096 candidates");
    ARL_REQUIRE(!notify_or_null || (reinterpret_cast<uintptr_t>(notify_or_null) & 7u) == 0, ARL_E_ALIGN, "notify: 8-byte aligned");
    hipLaunchKernelGGL(sumtree_sample_kernel, dim3(1), dim3(sample_threads(m)), 0, (hipStream_t)stream, tree, levels, uniforms,
                       m, n, part_size, tree_idxs, env_idxs, step_idxs, probs, n_unique, beta, is_weights_or_null,
                       (volatile long long*)notify_or_null, (long long)ticket);
    return arl::check_launch("sumtree_sample_kernel");
}

extern "C" int arl_sumtree_update_pow(double* tree, int32_t levels, const int32_t* tree_idxs, const float* priorities,
                                      const double* last_probs, double alpha, int64_t n, void* stream) {
    ARL_REQUIRE(tree && tree_idxs && priorities && last_probs, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(levels >= 1 && levels <= 31 && n >= 0, ARL_E_RANGE, "bad levels / n");
    for (int64_t lo = 0; lo < n; lo += ADD_MAX) {            // (consecutive chunks are exact: arl_sumtree_add)
        const int m = (int)((n - lo < ADD_MAX) ? n - lo : ADD_MAX);
        hipLaunchKernelGGL(sumtree_add_kernel, dim3((unsigned)levels), dim3(1024), 0, (hipStream_t)stream, tree,
                           tree_idxs + lo, (const double*)nullptr, m, priorities + lo, last_probs + lo, alpha);
        int rc = arl::check_launch("sumtree_add_kernel");
        if (rc) return rc;
    }
    return 0;
}

extern "C" int arl_is_weights(const double* probs, int64_t n, double beta, float* out, void* stream) {
    ARL_REQUIRE(probs && out, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(n >= 1 && n < ((int64_t)1 << 31), ARL_E_RANGE, "bad n");
    hipLaunchKernelGGL(is_weights_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, probs, (int)n, beta, out);
    return arl::check_launch("is_weights_kernel");
}

extern "C" int arl_priority_diffs(const float* priorities, const double* last_probs, int64_t n, double alpha,
                                  double* diffs, void* stream) {
    ARL_REQUIRE(priorities && last_probs && diffs, ARL_E_ARG, "null pointer");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(priority_diffs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       priorities, last_probs, n, alpha, diffs);
    return arl::check_launch("priority_diffs_kernel");
}

extern "C" int arl_sumtree_gather(const double* tree, const int32_t* tree_idxs, int64_t n, double scale,
                                  double* out, void* stream) {
    ARL_REQUIRE(tree && tree_idxs && out, ARL_E_ARG, "null pointer");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(sumtree_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       tree, tree_idxs, n, scale, out);
    return arl::check_launch("sumtree_gather_kernel");
}

extern "C" int arl_sumtree_add(double* tree, int32_t levels, const int32_t* tree_idxs, const double* diffs,
                               int64_t n, void* stream) {
    ARL_REQUIRE(tree && tree_idxs && diffs, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(levels >= 1 && levels <= 31 && n >= 0, ARL_E_RANGE, "bad levels / n");
    // consecutive chunks are exact: levels touch disjoint nodes and a node sees chunk A's updates before B's
    for (int64_t lo = 0; lo < n; lo += ADD_MAX) {
        const int m = (int)((n - lo < ADD_MAX) ? n - lo : ADD_MAX);
        hipLaunchKernelGGL(sumtree_add_kernel, dim3((unsigned)levels), dim3(1024), 0, (hipStream_t)stream, tree,
                           tree_idxs + lo, diffs + lo, m);
        int rc = arl::check_launch("sumtree_add_kernel");
        if (rc) return rc;
    }
    return 0;
}
