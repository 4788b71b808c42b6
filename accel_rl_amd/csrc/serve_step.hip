// One agent step of action serving as ONE launch (arl_env_step_served): the policy's output layers, the action draw, the
// env step and the next observation's first convolution, one workgroup of 16 waves per env.
//
// Reference being replaced (paths under the reference root): one (step, group) turn of
//   accel_rl/sampler/act_server/alternating/overlap/sampler.py:120-151   serve_actions: policy outputs -> sampled actions
//                                                                        -> scatter to the workers
//   accel_rl/policies/pg/atari_cnn_policy.py:63-67, pg/networks/pg_cnn.py:57-86   hidden layer's bias + rectifier, output layers
//   rllab/misc/special.py:22-27                                          weighted_sample_n
//   accel_rl/sampler/act_server/alternating/overlap/worker.py:37-59      ResetCollector: env.step, TrajInfo, reset, obs write
//   accel_rl/envs/atari_env.py:65-78,93-100,151-191                      AtariEnv.step / reset / _update_obs
//   accel_rl/policies/pg/networks/pg_cnn.py:47-52                        first Conv2DLayer on the NEXT observation
//
// Why one launch: at the rollout's 256 envs every launch of the per-step chain is latency, not work (fold of the hidden
// layer's split partials 6.6 us, heads 4.7 us, env step 13 us, conv 1 11.9 us of a 95 us step: profiles/r06/
// step_timeline.txt), and all four are "one workgroup per env" shaped.  Here the env's frame loads, the hidden layer's
// partial sums and conv 1's weights are all in flight together: which bank frames a step pushes does not depend on the
// sampled action (the emulator's tick advances whatever the action; the action only enters the reward), so the frame plan
// is derived BEFORE the heads are evaluated and the pixel work runs under them.
//
// Arithmetic: every stage repeats the unfused kernels' operations in their order -- fold_splits_kernel's two-level sum
// (mfma_conv.hip), head_kernel<false>'s dot products, butterfly sums, softmax (learner.hip), sample_action / step_compute /
// step_commit (env_dev.h), conv1_tile (img_conv_dev.h) -- so a served rollout is bit-identical to
// arl_fold + arl_pg_head_infer + arl_env_step(single_write) + arl_conv2d_u8_fwd (tests/test_serve_step_gpu.py).
#include "img_conv_dev.h"
#include "env_dev.h"
#include "head_dev.h"

namespace arlc {
// img_conv.hip: conv1_img_kernel with the copy-out (>= 0: launched / error code; -1: not its geometry or route)
int launch_conv1_img_begin(const arl_conv_geom* geom, const unsigned char* obs, int64_t n_img, float scale, const float* w,
                           const float* bias, float* y, int C, int relu, hipStream_t s, unsigned char* copy_out,
                           long long copy_stride, int* zero_word);
int fold_wide_from();           // mfma_conv.hip: the split count from which a fold sums 64-way (arl_dev_fold_wide_from)
extern bool g_no_img_kernels;   // img_conv.hip (arl_dev_conv_variant)
}

namespace {

constexpr int SV_NW = 16, SV_NT = SV_NW * 64;
constexpr int SV_HID_MAX = 1024;
constexpr int SV_FOLD_LDS_MAX = 64 * 1024;

struct ServeHead {
    const float4* part;         // [splits][n_env][hid / 4] partial sums of the last hidden layer (splits == 0: finished)
    const float4* bias;         // [hid / 4] or null
    const float* w_head;        // [A + 1][hid]
    const float* b_head;        // [A + 1]
    int64_t total4;             // n_env * hid / 4
    int splits, zgn, hid, relu;
};

// Order of work (what overlaps what):
//   top      every wave: conv 1's weight fragment and (waves with a head row) that row of W_head start their way from L2
//   wave 0   the frame plan: the stream's reset ranks (wave ballots), the env's state, step_compute -- lane j for
//            action j: which frames a step pushes, whether it ends the episode and the state it leaves do not depend on
//            the action (the emulator's tick advances whatever it is; the action only enters the reward), so the plan
//            is known before the heads are evaluated and the lane of the action drawn later simply commits ITS outcome
//   waves 1+ meanwhile: the hidden layer's split partial sums, fold_splits_kernel's first level
//   sync A   -> frame loads issued (they fly under everything below), weights split into LDS, arrival ticket;
//            fold's second level + bias + rectifier
//   sync B   -> head rows (one per wave, butterfly sums); pixels boxed and stored (rollout buffer + LDS)
//   sync C   -> wave 0: softmax, the draw, commit;  waves 1+: conv 1 of the new observation from LDS
template <bool CONV1>
__global__ __launch_bounds__(SV_NT) void serve_step_kernel(
    const arl_game g, const arl_env_state st, const arl_rollout ro, const ServeHead hd, const arlc::Conv1ImgArgs c1,
    const double* __restrict__ uniforms, int step, double max_path_length, double discount, int max_start_noops) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ int s_plan[4];                             // fa, fb, mode
    __shared__ __attribute__((aligned(16))) float s_h[SV_HID_MAX];
    __shared__ float s_o[32];
    const int64_t e = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nsteps = CONV1 ? c1.C * 4 : 0;
    char* const sW = lds;                                 // conv 1's weights, [nsteps][3 planes][64 lanes] x 16 bytes
    char* const sI = lds + nsteps * 3072;                 // the fold's scratch, later the new observation [F][104][80]
    const int A = g.n_actions, K = A + 1, hid = hd.hid, hid4 = hid >> 2;

    // ---- top: loads that depend on nothing
    arlc::Conv1WFrag wf = {};
    const bool has_w = CONV1 && tid < nsteps * 64;
    if (has_w) wf = arlc::conv1_w_load(c1.w, c1.C * 64, tid, c1.NF);
    // head row k goes to wave (k + 1) % 16: wave 0, which walks the step's serial part, gets one only from 16 rows on
    const int k_first = (wave + SV_NW - 1) % SV_NW;
    float wrow[SV_HID_MAX / 64];
    float brow = 0.f;
    if (k_first < K) {
#pragma unroll
        for (int j = 0; j < SV_HID_MAX / 64; ++j) wrow[j] = lane + 64 * j < hid ? hd.w_head[k_first * hid + lane + 64 * j] : 0.f;
        brow = hd.b_head[k_first];
    }
    const int parity = st.epoch[0] & 1;
    const int fpar = st.launch_count[0] & 1;              // this state's own launch parity (the epoch is shared)

    StepOut o = {};
    double u = 0.0;
    int64_t* lead_cursor = nullptr;
    int64_t lead_value = 0;
    bool mismatch = false;
    float4* sF = reinterpret_cast<float4*>(sI);
    if (wave == 0) {
        // ---- the frame plan (env_step_kernel, steps (1) and (2)) on one wave
        u = uniforms[e];
        const int64_t per = st.envs_per_stream;
        const int64_t w = e / per, g0 = w * per;
        const int64_t hi = g0 + per < st.n_env ? g0 + per : st.n_env;
        const int64_t n_streams = (st.n_env + per - 1) / per;
        const int64_t cursor = st.noop_cursor[parity * n_streams + w];
        const uint8_t* flag_now = st.next_reset + (int64_t)fpar * st.n_env;         // written by the previous launch
        const EnvRegs in = load_env(st, e);
        const uint8_t carried = flag_now[e];
        int rank = 0, total = 0;
        if (max_start_noops > 0) {                        // (no draws otherwise: nothing to rank)
            for (int64_t base = g0; base < hi; base += 64) {
                const int64_t i = base + lane;
                const bool f = i < hi && flag_now[i] != 0;
                rank += __popcll(__ballot(f && i < e));
                total += __popcll(__ballot(f));
            }
        }
        o = step_compute(g, in, lane < A ? lane : 0, true, 1, max_path_length, discount);
        if (o.reset_flag) {                               // env.reset() (worker.py:47): start no-ops from the stream's ring
            int noops = 0;
            if (max_start_noops > 0) noops = st.noop_ring[w * st.noop_ring_len + (cursor + rank) % st.noop_ring_len];
            reset_regs(g, o.s, noops, o.fa, o.fb, o.mode);
        }
        if (lane == 0) { s_plan[0] = o.fa; s_plan[1] = o.fb; s_plan[2] = o.mode; }
        lead_cursor = e == g0 ? st.noop_cursor + (parity ^ 1) * n_streams + w : nullptr;
        lead_value = cursor + total;
        mismatch = (o.reset_flag != 0) != (carried != 0);
    } else if (hd.splits > 0) {
        // ---- the hidden layer, first level of fold_splits_kernel's sum for this env's row: zgn threads share an output
        // float4, thread zg sums splits zg, zg + zgn, ...
        const float4* part = hd.part + e * hid4;
        for (int it = tid - 64; it < hid4 * hd.zgn; it += SV_NT - 64) {
            const int zg = it / hid4, c4 = it - zg * hid4;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int z = zg; z < hd.splits; z += hd.zgn) {
                const float4 v = part[(int64_t)z * hd.total4 + c4];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            sF[it] = s;
        }
    }
    __syncthreads();                                      // ---- A: the plan and the first-level sums are out
    const int fa = s_plan[0], fb = s_plan[1], mode = s_plan[2];
    const int64_t row_bytes = (int64_t)g.n_stack * OBS_FRAME;
    uint8_t* next = step + 1 < ro.horizon ? ro.observations + (e * ro.horizon + step + 1) * row_bytes : nullptr;
    const uint8_t* prev = ro.observations + (e * ro.horizon + step) * row_bytes;
    uint8_t* out0 = next ? next : ro.step_obs + e * row_bytes;
    FramePush<true, SV_NT> fp;
    fp.template load<0>(g, fa, fb, mode, prev, tid);       // every load of the pixel work is in flight from here
    if (has_w) arlc::conv1_w_store(sW, tid, wf);          // (the weights have arrived long since: split, into LDS)
    if (tid == 0) {
        // the step's stores that do not depend on the action (env_step_kernel: the stream leader's next cursor, the reset
        // forecast for the next launch and the check of this launch's)
        if (lead_cursor) *lead_cursor = lead_value;
        st.next_reset[(int64_t)(fpar ^ 1) * st.n_env + e] = (uint8_t)will_reset(g, o.s, max_path_length);
        if (mismatch) atomicAdd(st.epoch + 2, 1);
    }
    if (tid == SV_NT - 1) {
        // Arrival ticket (env_step_kernel, step (4)): every thread of this workgroup has read both counters.  Two levels:
        // env e arrives on shard e % 16, a shard's last arriver on the top word, the last of those bumps the counters.
        const int n = (int)st.n_env, sh = (int)(e % ARL_TICKET_SHARDS);
        const int expect = (n - sh + ARL_TICKET_SHARDS - 1) / ARL_TICKET_SHARDS;
        int* shard = st.epoch + 32 * (sh + 1);
        if (atomicAdd(shard, 1) == expect - 1) {
            *shard = 0;
            const int shards = n < ARL_TICKET_SHARDS ? n : ARL_TICKET_SHARDS;
            if (atomicAdd(st.epoch + 1, 1) == shards - 1) {
                st.epoch[1] = 0;
                st.epoch[0] += 1;
                st.launch_count[0] += 1;
            }
        }
    }
    // ---- second level: the zgn sums in index order, bias, rectifier (threads 64 .. 64 + hid / 4)
    if (tid >= 64 && tid - 64 < hid4) {
        const int c4 = tid - 64;
        float4 s;
        if (hd.splits > 0) {
            s = sF[c4];
            for (int k = 1; k < hd.zgn; ++k) {
                const float4 v = sF[k * hid4 + c4];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            if (hd.bias) {
                const float4 b = hd.bias[c4];
                s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
            }
            if (hd.relu) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
        } else {
            s = (hd.part + e * hid4)[c4];                 // finished activations
        }
        reinterpret_cast<float4*>(s_h)[c4] = s;
    }
    __syncthreads();                                      // ---- B: the hidden activations are out (and sF is free)

    // ---- output layers: head_kernel's dot products (lanes split the hidden dimension, j ascending; butterfly sum)
    for (int k = k_first; k < K; k += SV_NW) {
        float sdot = 0.f;
        if (k == k_first) {
#pragma unroll
            for (int j = 0; j < SV_HID_MAX / 64; ++j)
                if (lane + 64 * j < hid) sdot += s_h[lane + 64 * j] * wrow[j];
        } else {
            for (int c = lane; c < hid; c += 64) sdot += s_h[c] * hd.w_head[k * hid + c];
            brow = hd.b_head[k];
        }
        const float ov = wave_sum_f(sdot) + brow;
        if (lane == 0) s_o[k] = ov;
    }
    // ---- the pixels: new frame = rounded 2x2 box of the cropped max of the two bank frames, older planes shifted; to
    // the rollout buffer (step_obs after the batch's last step) and, for conv 1, to LDS
    fp.template store<0>(g, out0, nullptr, tid, CONV1 ? reinterpret_cast<uint8_t*>(sI) : nullptr);
    __syncthreads();                                      // ---- C: head outputs and the observation are out

    if (wave == 0) {
        // ---- softmax, the action draw, and the drawn action's lane commits its outcome (head_kernel<false>,
        // sample_action, step_commit)
        float v = 0.f, mx = -3.0e38f, mine = 0.f;
        for (int k = 0; k < K; ++k) {
            const float ok = s_o[k];
            if (k < A) mx = fmaxf(mx, ok); else v = ok;
            mine = (lane == k) ? ok : mine;
        }
        const bool is_act = lane < A;
        const float ex = is_act ? expf(mine - mx) : 0.f;
        float z = 0.f;
        for (int k = 0; k < A; ++k) z += readlane_f(ex, k);
        const float pk = ex / z;
        const int64_t row = e * ro.horizon + step;
        if (is_act) ro.prob[row * A + lane] = pk;
        float c = 0.f;
        int kk = 0;
        for (int j = 0; j < A; ++j) {                         // weighted_sample_n: fp32 sequential cumsum, f64 compare
            c += readlane_f(pk, j);
            kk += ((double)c < u) ? 1 : 0;
        }
        const int a_idx = kk < A - 1 ? kk : A - 1;
        if (lane == a_idx) step_commit<false>(g, st, ro, nullptr, v, o, e, step);
    }
    if (CONV1) {
        // ---- conv 1 of the observation just written, from LDS: the 32-pixel row tiles go to waves 1 .. 15 first (wave 0
        // is busy above); bit for bit conv1_img_kernel's tiles
        float4 bq[4];
        arlc::conv1_bias_quads(c1.bias, lane >> 5, bq, c1.NF);
        const int tiles = (c1.OH * c1.OW + 31) / 32;
        for (int tp = (wave + SV_NW - 1) % SV_NW; tp < tiles; tp += SV_NW) arlc::conv1_tile(c1, sI, sW, (int)e, tp, lane, bq);
    }
}

}  // namespace

extern "C" int arl_serve_conv1_supported(const arl_game* game, const arl_conv_geom* geom) {
    if (!game || !geom) return 0;
    const int64_t npix = (int64_t)game->n_stack * OBS_FRAME;
    return geom->in_c == game->n_stack && geom->in_h == ARL_OBS_H && geom->in_w == ARL_OBS_W && (geom->out_c == 32 || geom->out_c == 16) &&
           geom->kh == 8 && geom->kw == 8 && geom->pad_h == 0 && geom->pad_w == 0 && geom->stride > 0 &&
           (geom->stride & 3) == 0 && geom->stride <= 8 && game->n_stack * 64 <= SV_NT && npix <= arlc::C1_MAX_IMG &&
           (geom->route == ARL_CONV_ROUTE_SPLIT9 || geom->route == ARL_CONV_ROUTE_SPLIT6) && !arlc::g_no_img_kernels;
}

extern "C" int arl_env_step_served(const arl_game* game, const arl_env_state* st, const arl_rollout* ro,
                                   const arl_serve_head* head, const arl_serve_conv1* conv1_or_null,
                                   const double* uniforms, int32_t step, double max_path_length, double discount,
                                   int32_t max_start_noops, void* stream) {
    int rc = check_env_args(game, st, ro);
    if (rc) return rc;
    ARL_REQUIRE(head && uniforms, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(ro->rewards && ro->dones && ro->actions && ro->prob && ro->value && ro->observations, ARL_E_ARG,
                "null rollout arrays");
    ARL_REQUIRE(step >= 0 && step < ro->horizon, ARL_E_RANGE, "step outside horizon");
    ARL_REQUIRE(arl::aligned16(ro->observations) && arl::aligned16(ro->step_obs), ARL_E_ALIGN,
                "observations and step_obs must be 16-byte aligned");
    ARL_REQUIRE(st->next_reset && st->launch_count, ARL_E_ARG, "needs st->next_reset and st->launch_count");
    ARL_REQUIRE(max_path_length >= 1.0, ARL_E_RANGE, "needs max_path_length >= 1");
    ARL_REQUIRE(game->n_stack <= MAX_STACK, ARL_E_RANGE, "frame stacks deeper than 4: use arl_env_step");
    // ---- the output layers' input
    const arl_fold_item& h = head->hidden;
    const int hid = head->hid;
    ARL_REQUIRE(h.part && head->w_head && head->b_head, ARL_E_ARG, "null head pointers");
    ARL_REQUIRE(hid > 0 && (hid & 3) == 0 && hid <= SV_HID_MAX, ARL_E_RANGE, "hid must be a multiple of 4, <= 1024");
    ARL_REQUIRE(h.total == st->n_env * (int64_t)hid && h.splits >= 0, ARL_E_ARG,
                "head->hidden must describe n_env rows of hid floats");
    ARL_REQUIRE(arl::aligned16(h.part) && (!head->hidden_bias || arl::aligned16(head->hidden_bias)), ARL_E_ALIGN,
                "16-byte alignment");
    ServeHead hd = {};
    hd.part = (const float4*)h.part; hd.bias = (const float4*)head->hidden_bias;
    hd.w_head = head->w_head; hd.b_head = head->b_head;
    hd.total4 = h.total >> 2; hd.splits = h.splits; hd.hid = hid; hd.relu = head->hidden_relu;
    hd.zgn = h.splits >= arlc::fold_wide_from() ? 64 : 16;              // fold_slot (mfma_conv.hip)
    const size_t fold_lds = h.splits > 0 ? (size_t)hd.zgn * hid * 4 : 0;
    ARL_REQUIRE(fold_lds <= SV_FOLD_LDS_MAX, ARL_E_RANGE, "hidden layer too wide for this many split partials");
    // ---- conv 1 of the next observation (optional)
    arlc::Conv1ImgArgs c1 = {};
    size_t lds = fold_lds;
    if (conv1_or_null) {
        const arl_serve_conv1& c = *conv1_or_null;
        ARL_REQUIRE(c.w && c.y && c.geom, ARL_E_ARG, "null conv1 pointers");
        ARL_REQUIRE(arl_serve_conv1_supported(game, c.geom), ARL_E_RANGE,
                    "conv1 geometry / route not served in the step launch (arl_serve_conv1_supported)");
        ARL_REQUIRE(c.geom->batch == st->n_env, ARL_E_ARG, "conv1 geometry must be built for n_env images");
        ARL_REQUIRE(arl::aligned16(c.w) && arl::aligned16(c.y) && (!c.bias || arl::aligned16(c.bias)), ARL_E_ALIGN,
                    "16-byte alignment");
        const int npix = game->n_stack * OBS_FRAME;
        c1.w = c.w; c1.bias = c.bias; c1.y = c.y; c1.scale = c.scale; c1.n_img = (int)st->n_env; c1.C = game->n_stack;
        c1.H = ARL_OBS_H; c1.W = ARL_OBS_W; c1.stride = c.geom->stride; c1.relu = c.relu; c1.NF = c.geom->out_c;
        c1.OH = (ARL_OBS_H - 8) / c.geom->stride + 1; c1.OW = (ARL_OBS_W - 8) / c.geom->stride + 1;
        lds = (size_t)game->n_stack * 4 * 3072 + (fold_lds > (size_t)npix ? fold_lds : (size_t)npix);
    }
    hipStream_t s = (hipStream_t)stream;
#define ARL_SERVE(C1_)                                                                                              \
    do {                                                                                                            \
        auto k = serve_step_kernel<C1_>;                                                                            \
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                             152 * 1024);                                                           \
        if (err != hipSuccess) { arl::set_error("hipFuncSetAttribute(serve_step_kernel): %s", hipGetErrorString(err)); return (int)err; } \
        hipLaunchKernelGGL(k, dim3((unsigned)st->n_env), dim3(SV_NT), lds, s, *game, *st, *ro, hd, c1, uniforms,     \
                           (int)step, max_path_length, discount, (int)max_start_noops);                             \
    } while (0)
    if (conv1_or_null) ARL_SERVE(true); else ARL_SERVE(false);
#undef ARL_SERVE
    return arl::check_launch("serve_step_kernel");
}

extern "C" int arl_rollout_begin_conv1(const arl_game* game, const arl_env_state* st, const arl_rollout* ro,
                                       const arl_serve_conv1* conv1, void* stream) {
    int rc = check_env_args(game, st, ro);
    if (rc) return rc;
    ARL_REQUIRE(ro->observations && conv1 && conv1->w && conv1->y && conv1->geom, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(arl_serve_conv1_supported(game, conv1->geom), ARL_E_RANGE,
                "conv1 geometry / route not served (arl_serve_conv1_supported): arl_rollout_begin + arl_conv2d_u8_fwd");
    ARL_REQUIRE(conv1->geom->batch == st->n_env, ARL_E_ARG, "conv1 geometry must be built for n_env images");
    ARL_REQUIRE(arl::aligned16(ro->step_obs) && arl::aligned16(ro->observations) && arl::aligned16(conv1->w) &&
                    arl::aligned16(conv1->y) && (!conv1->bias || arl::aligned16(conv1->bias)), ARL_E_ALIGN, "16-byte alignment");
    const int64_t row = (int64_t)game->n_stack * OBS_FRAME;
    rc = arlc::launch_conv1_img_begin(conv1->geom, ro->step_obs, st->n_env, conv1->scale, conv1->w, conv1->bias, conv1->y,
                                      game->n_stack, conv1->relu, (hipStream_t)stream, ro->observations,
                                      (long long)ro->horizon * row, st->done_count);
    if (rc < 0) { arl::set_error("arl_rollout_begin_conv1: the image kernel refused this geometry"); return ARL_E_RANGE; }
    return rc;
}
