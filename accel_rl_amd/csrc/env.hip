// Vectorised environment step for gfx950: synthetic fixed-frame emulator +
// AtariEnv wrapper + collector bookkeeping, all N envs of a learner per launch.
//
//   arl_env_act_step    one lane per env   (scalar rules; bit-exact integer logic)
//   arl_env_frame_step  one workgroup/env  (pixels: max, crop, 2x2 box, stack, store)
//   arl_env_reset       frame_step in "reset the flagged envs" mode
//
// Reference being replaced (paths under the reference root):
//   accel_rl/envs/atari_env.py:65-78,93-100,151-191      AtariEnv.step/reset/_update_obs/...
//   accel_rl/sampler/act_server/alternating/overlap/worker.py:25-113  collectors
//   accel_rl/sampler/act_server/alternating/overlap/sampler.py:139-145 action scatter
//   accel_rl/sampler/util.py:75-101                        TrajInfo
//   rllab/misc/special.py:22-27                            weighted_sample_n
// Emulator: oracle/synth_ale.py is the specification (ALE itself is third-party).
//
// Memory behaviour of frame_step (the dominant kernel of the rollout): per env
// it reads two 33 600-B raw frames (16-B loads, fully coalesced rows), reads
// the previous stack and writes the new stacked observation once to step_obs
// and once to the env-major rollout buffer (8-B stores, 80-B rows).  HBM-bound.

#include "env_dev.h"

namespace {

__global__ __launch_bounds__(256) void act_step_kernel(
    const arl_game g, const arl_env_state st, const arl_rollout ro, const float* __restrict__ prob,
    const float* __restrict__ value, const double* __restrict__ uniforms,
    const uint8_t* __restrict__ active, int step, int mid_batch_reset, double max_path_length,
    double discount) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= st.n_env) return;
    const EnvRegs in = load_env(st, e);
    const float* p = prob + e * g.n_actions;
    const StepOut o = step_compute(g, in, sample_action(p, g.n_actions, uniforms[e]), !active || active[e] != 0,
                                   mid_batch_reset, max_path_length, discount);
    step_commit(g, st, ro, p, value[e], o, e, step);
}


// RESET_ONLY: flags come from the caller (NULL = all), only step_obs is written.
template <bool RESET_ONLY>
__global__ __launch_bounds__(256) void frame_step_kernel(const arl_game g, const arl_env_state st,
                                                         const arl_rollout ro,
                                                         const uint8_t* __restrict__ ext_flags,
                                                         int step, int max_start_noops) {
    const int64_t e = blockIdx.x;
    const int tid = threadIdx.x;
    __shared__ int s_fa, s_fb, s_mode;
    const int parity = st.epoch[0] & 1;
    const int64_t per = st.envs_per_stream;
    const uint8_t* flags = RESET_ONLY ? ext_flags : st.reset_flag;
    const bool flagged = RESET_ONLY ? (ext_flags ? ext_flags[e] != 0 : true) : (st.reset_flag[e] != 0);

    if (tid == 0) {
        if (flagged) resolve_reset(g, st, e, RESET_ONLY && !ext_flags ? nullptr : flags, max_start_noops, parity);
        else if (RESET_ONLY) st.frame_mode[e] = MODE_SKIP;
        s_fa = st.frame_a[e]; s_fb = st.frame_b[e]; s_mode = st.frame_mode[e];
        if (e % per == 0) {                               // stream leader: publish next cursor
            const int64_t w = e / per, n_streams = (st.n_env + per - 1) / per;
            int64_t hi = e + per < st.n_env ? e + per : st.n_env;
            int total = 0;
            if (max_start_noops > 0)
                for (int64_t i = e; i < hi; ++i) total += (RESET_ONLY && !ext_flags) ? 1 : (flags[i] != 0);
            st.noop_cursor[(parity ^ 1) * n_streams + w] = st.noop_cursor[parity * n_streams + w] + total;
        }
        if (RESET_ONLY && flagged) st.frozen[e] = 0;
        // arl_env_step's reset forecast (will_reset): a freshly reset env starts a trajectory (Length 0, so no
        // over-length at the limits >= 1 that arl_env_step accepts); the others keep theirs
        if (RESET_ONLY && st.next_reset) {
            const int fpar = st.launch_count[0] & 1;
            st.next_reset[(int64_t)(fpar ^ 1) * st.n_env + e] =
                flagged ? (uint8_t)will_reset(g, load_env(st, e), 1e300) : st.next_reset[(int64_t)fpar * st.n_env + e];
        }
    }
    __syncthreads();
    if (s_mode == MODE_SKIP) return;
    uint8_t* cur = ro.step_obs + e * (int64_t)g.n_stack * OBS_FRAME;
    uint8_t* dst = nullptr;
    if (!RESET_ONLY && step + 1 < ro.horizon)             // worker.py:52-53
        dst = ro.observations + (e * ro.horizon + step + 1) * (int64_t)g.n_stack * OBS_FRAME;
    push_frame(g, s_fa, s_fb, s_mode, cur, cur, dst, tid);
}

// The whole env side of one agent step in ONE launch, one workgroup per env:
//   (1) every lane evaluates will_reset for one env of this env's worker stream -> this env's rank among the
//       stream's resets and (stream leader) their number, i.e. what frame_step used to read from the flags
//       act_step had stored one launch earlier;
//   (2) lane 0 runs act_one (sampling, scatter, emulator, rules, TrajInfo), resolves a pending reset and, as
//       stream leader, publishes the stream's next no-op cursor;
//   (3) the workgroup pushes the new frame (push_frame);
//   (4) the last workgroup to arrive bumps the launch epoch (st.epoch[1] is the arrival ticket), so no
//       epoch_kernel launch follows.
// single_write (needs mid_batch_reset: no env sits a step out): the stacked observation is written ONCE -- to
// observations[e][step + 1], or to step_obs after the batch's last step -- and the previous stack is read from
// observations[e][step]; the policy then reads the current observations as rows e * horizon + step of the
// rollout buffer instead of step_obs.  Otherwise step_obs is kept current at every step (second write).
// KO (development, arl_dev_env_variant): knock-outs for timing only (results wrong) -- 1: every env reads bank frame 0;
// 2: the older planes of the stack are not stored; 3: ... nor loaded.
// SW == (single_write != 0), the launcher's dispatch: the wide copy of the older planes (FramePush<true>).
template <int KO, bool SW>
__global__ __launch_bounds__(256) void env_step_kernel(
    const arl_game g, const arl_env_state st, const arl_rollout ro, const float* __restrict__ prob,
    const float* __restrict__ value, const double* __restrict__ uniforms,
    const uint8_t* __restrict__ active, int step, int mid_batch_reset, double max_path_length,
    double discount, int max_start_noops, int single_write) {
    const int64_t e = blockIdx.x;
    const int tid = threadIdx.x;
    const int parity = st.epoch[0] & 1;
    const int fpar = st.launch_count[0] & 1;              // this state's own launch parity (the epoch is shared)
    const int64_t per = st.envs_per_stream;
    const int64_t w = e / per, g0 = w * per;
    const int64_t hi = g0 + per < st.n_env ? g0 + per : st.n_env;
    const int64_t n_streams = (st.n_env + per - 1) / per;
    // ---- wave 0 derives the step from the env's state, the served distribution and the uniform (uniform addresses:
    // broadcast loads) and hands the frame plan to the other three waves through LDS.  (Rounds 2-3 had EVERY lane derive
    // it -- no hand-off, but ~270 of a wave's ~870 vector instructions, a quarter of the launch's issue time at 16 384
    // envs: tools/env_step_bound.sh.)
    __shared__ int s_plan[4];                             // fa, fb, mode, stepped
    const bool planner = tid < 64;                        // (wave-uniform)
    const float* p = prob + e * g.n_actions;
    const bool is_active = !active || active[e] != 0;
    const int64_t cursor = st.noop_cursor[parity * n_streams + w];
    const uint8_t* flag_now = st.next_reset + (int64_t)fpar * st.n_env;         // written by the previous launch
    int rank = 0, total = 0;
    if (mid_batch_reset && max_start_noops > 0) {         // (no draws otherwise: nothing to rank)
        for (int64_t base = g0; base < hi; base += 256) {
            const int64_t i = base + tid;
            const bool f = i < hi && flag_now[i] != 0 && (!active || active[i] != 0);
            rank += __syncthreads_count(f && i < e);
            total += __syncthreads_count(f);
        }
    }
    const uint8_t carried = flag_now[e];
    StepOut o = {};
    float v = 0.f;
    if (planner) {
        const EnvRegs in = load_env(st, e);
        const int a_idx = sample_action(p, g.n_actions, uniforms[e]);
        v = value[e];
        o = step_compute(g, in, a_idx, is_active, mid_batch_reset, max_path_length, discount);
        if (o.reset_flag) {                               // env.reset() (worker.py:47): start no-ops from the stream's ring
            int noops = 0;
            if (max_start_noops > 0)                      // randint(0, 1) draws nothing
                noops = st.noop_ring[w * st.noop_ring_len + (cursor + rank) % st.noop_ring_len];
            reset_regs(g, o.s, noops, o.fa, o.fb, o.mode);
        }
        if (tid == 0) { s_plan[0] = o.fa; s_plan[1] = o.fb; s_plan[2] = o.mode; s_plan[3] = o.stepped; }
    }
    __syncthreads();                                      // the plan is out (and every wave-0 lane has read the state
    //                                                       lane 0 is about to overwrite)
    o.fa = s_plan[0]; o.fb = s_plan[1]; o.mode = s_plan[2]; o.stepped = s_plan[3];
    const int64_t row_bytes = (int64_t)g.n_stack * OBS_FRAME;
    uint8_t* cur = ro.step_obs + e * row_bytes;
    uint8_t* next = step + 1 < ro.horizon ? ro.observations + (e * ro.horizon + step + 1) * row_bytes : nullptr;
    const uint8_t* prev = single_write ? ro.observations + (e * ro.horizon + step) * row_bytes : cur;
    uint8_t* out0 = single_write ? (next ? next : cur) : cur;
    uint8_t* out1 = single_write ? nullptr : next;            // worker.py:52-53
    const bool fast = g.n_stack <= MAX_STACK;
    FramePush<SW> fp;                                     // SW == (single_write != 0): the launcher's dispatch
    if (KO == 1) { o.fa = o.fa >= 0 ? 0 : -1; o.fb = 0; }
    if (o.mode != MODE_SKIP && fast) fp.template load<KO>(g, o.fa, o.fb, o.mode, prev, tid);
    // the stores of the scalar part, under the frame loads' latency: lane 0 commits, lanes of wave 1 copy the
    // served distribution (one element each instead of a load -> store chain per action on lane 0)
    if (o.stepped && tid >= 64 && tid - 64 < g.n_actions)
        ro.prob[(e * ro.horizon + step) * g.n_actions + (tid - 64)] = p[tid - 64];
    if (tid == 0) {
        step_commit<false>(g, st, ro, p, v, o, e, step);
        if (e == g0) st.noop_cursor[(parity ^ 1) * n_streams + w] = cursor + total;   // stream leader: next cursor
        // this env's flag for the next launch: recomputed if it stepped, carried over if it sat this one out
        st.next_reset[(int64_t)(fpar ^ 1) * st.n_env + e] = is_active ? (uint8_t)will_reset(g, o.s, max_path_length)
                                                                     : carried;
        // the ranks above came from the PREVIOUS launch's forecast, the reset itself from this launch's rules: they
        // agree as long as termination does not depend on the action and consecutive launches use one length limit.
        // A disagreement would misorder the streams' no-op draws silently -- count it instead (epoch[2], sticky; the
        // sampler raises when its mirror of the block shows a non-zero count)
        if (mid_batch_reset && is_active && (o.reset_flag != 0) != (carried != 0)) atomicAdd(st.epoch + 2, 1);
    }
    if (o.mode != MODE_SKIP) {
        if (fast) fp.template store<KO>(g, out0, out1, tid);
        else push_frame(g, o.fa, o.fb, o.mode, prev, out0, out1, tid);
    }
    if (tid == 0) {
        // Arrival ticket, two levels (256 arrivals on ONE word cost ~3 us, ~12 ns each, serialised at its L2
        // channel): env e arrives on shard e % 16 (own 128-byte line each); a shard's last arriver arrives on the
        // top word; the last of those bumps the counters.  Everyone has read both counters before arriving.
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
}

// bump the launch epoch AFTER a frame_step launch (single lane)
__global__ void epoch_kernel(int32_t* epoch, int32_t* launch_count) {
    epoch[0] += 1;
    if (launch_count) launch_count[0] += 1;
}

__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* __restrict__ a,
                                                         const uint8_t* __restrict__ b,
                                                         uint8_t* __restrict__ out, const int resample_mode) {
    const int64_t i = blockIdx.x;
    const uint8_t* fb = b + i * RAW_FRAME;
    const uint8_t* fa = a ? a + i * RAW_FRAME : nullptr;
    for (int un = threadIdx.x; un < UNITS; un += blockDim.x) {
        const int y = un / UNITS_PER_ROW, xb = un - y * UNITS_PER_ROW;
        const int src = (2 * y) * ARL_RAW_W + xb * 16;
        const uint4 b0 = *reinterpret_cast<const uint4*>(fb + src);
        const uint4 b1 = *reinterpret_cast<const uint4*>(fb + src + ARL_RAW_W);
        uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
        if (fa) {
            a0 = *reinterpret_cast<const uint4*>(fa + src);
            a1 = *reinterpret_cast<const uint4*>(fa + src + ARL_RAW_W);
        }
        *reinterpret_cast<uint2*>(out + i * OBS_FRAME + un * 8) = resample8(resample_mode, a0, a1, b0, b1);
    }
}

int g_env_variant = 0;               // arl_dev_env_variant: timing knock-outs of env_step_kernel (development)

int launch_env_step(const arl_game* game, const arl_env_state* st, const arl_rollout* ro, const float* prob,
                    const float* value, const double* uniforms, const uint8_t* active, int32_t step,
                    int32_t mid_batch_reset, double max_path_length, double discount, int32_t max_start_noops,
                    int32_t single_write, hipStream_t s) {
#define ARL_ENV_STEP(KO_, SW_)                                                                                   \
    hipLaunchKernelGGL((env_step_kernel<KO_, SW_>), dim3((unsigned)st->n_env), dim3(256), 0, s, *game, *st, *ro, \
                       prob, value, uniforms, active, (int)step, (int)mid_batch_reset, max_path_length, discount, \
                       (int)max_start_noops, (int)single_write)
    if (!single_write) ARL_ENV_STEP(0, false);            // (the timing knock-outs exist for the single-write kernel)
    else switch (g_env_variant) {
        case 1: ARL_ENV_STEP(1, true); break;
        case 2: ARL_ENV_STEP(2, true); break;
        case 3: ARL_ENV_STEP(3, true); break;
        default: ARL_ENV_STEP(0, true);
    }
#undef ARL_ENV_STEP
    return arl::check_launch("env_step_kernel");
}

}  // namespace

extern "C" int arl_env_act_step(const arl_game* game, const arl_env_state* st, const arl_rollout* ro,
                                const float* prob, const float* value, const double* uniforms,
                                const uint8_t* active_or_null, int32_t step,
                                int32_t mid_batch_reset, double max_path_length, double discount,
                                void* stream) {
    int rc = check_env_args(game, st, ro);
    if (rc) return rc;
    ARL_REQUIRE(prob && value && uniforms, ARL_E_ARG, "null policy outputs");
    ARL_REQUIRE(ro->rewards && ro->dones && ro->actions && ro->prob && ro->value, ARL_E_ARG, "null rollout arrays");
    ARL_REQUIRE(step >= 0 && step < ro->horizon, ARL_E_RANGE, "step outside horizon");
    const unsigned grid = (unsigned)((st->n_env + 255) / 256);
    hipLaunchKernelGGL(act_step_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *game, *st, *ro,
                       prob, value, uniforms, active_or_null, (int)step, (int)mid_batch_reset,
                       max_path_length, discount);
    return arl::check_launch("act_step_kernel");
}

extern "C" int arl_env_frame_step(const arl_game* game, const arl_env_state* st, const arl_rollout* ro,
                                  int32_t step, int32_t max_start_noops, void* stream) {
    int rc = check_env_args(game, st, ro);
    if (rc) return rc;
    ARL_REQUIRE(ro->observations, ARL_E_ARG, "null observations");
    ARL_REQUIRE(step >= 0 && step < ro->horizon, ARL_E_RANGE, "step outside horizon");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL((frame_step_kernel<false>), dim3((unsigned)st->n_env), dim3(256), 0, s, *game,
                       *st, *ro, (const uint8_t*)nullptr, (int)step, (int)max_start_noops);
    rc = arl::check_launch("frame_step_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(epoch_kernel, dim3(1), dim3(1), 0, s, st->epoch, (int32_t*)nullptr);
    return arl::check_launch("epoch_kernel");
}

extern "C" int arl_env_step(const arl_game* game, const arl_env_state* st, const arl_rollout* ro,
                            const float* prob, const float* value, const double* uniforms,
                            const uint8_t* active_or_null, int32_t step, int32_t mid_batch_reset,
                            double max_path_length, double discount, int32_t max_start_noops,
                            int32_t single_write, void* stream) {
    int rc = check_env_args(game, st, ro);
    if (rc) return rc;
    ARL_REQUIRE(prob && value && uniforms, ARL_E_ARG, "null policy outputs");
    ARL_REQUIRE(ro->rewards && ro->dones && ro->actions && ro->prob && ro->value && ro->observations, ARL_E_ARG,
                "null rollout arrays");
    ARL_REQUIRE(step >= 0 && step < ro->horizon, ARL_E_RANGE, "step outside horizon");
    ARL_REQUIRE(!single_write || (mid_batch_reset && !active_or_null), ARL_E_ARG,
                "single_write needs mid_batch_reset and every env stepping");
    // single_write moves the older planes 16 bytes per lane (FramePush<true>): rows of the rollout buffer are
    // 33 280-byte multiples, so 16-byte aligned bases make every access aligned
    ARL_REQUIRE(!single_write || (arl::aligned16(ro->observations) && arl::aligned16(ro->step_obs)),
                ARL_E_ALIGN, "single_write needs 16-byte aligned observations and step_obs");
    ARL_REQUIRE(st->next_reset && st->launch_count, ARL_E_ARG, "arl_env_step needs st->next_reset and st->launch_count");
    ARL_REQUIRE(max_path_length >= 1.0, ARL_E_RANGE, "arl_env_step needs max_path_length >= 1");
    return launch_env_step(game, st, ro, prob, value, uniforms, active_or_null, step, mid_batch_reset, max_path_length,
                           discount, max_start_noops, single_write, (hipStream_t)stream);
}

extern "C" void arl_dev_env_variant(int32_t v) { g_env_variant = (v >= 0 && v <= 3) ? v : 0; }

// Start of a batch: the current observation of every env becomes row (env, 0) of the rollout buffer (worker.py:30-32)
// and the completed-trajectory counter restarts -- one launch of 16-byte copies (as two framework launches, a strided
// element-wise copy and a fill, it was 9.6 + 4.9 us of the 256-env rollout).
__global__ __launch_bounds__(256) void batch_begin_kernel(const uint4* __restrict__ step_obs, uint4* __restrict__ observations,
                                                          int row16, int horizon, int32_t* done_count) {
    const int64_t e = blockIdx.x;
    const uint4* src = step_obs + e * row16;
    uint4* dst = observations + e * horizon * row16;
    for (int i = threadIdx.x; i < row16; i += 256) dst[i] = src[i];
    if (e == 0 && threadIdx.x == 0) done_count[0] = 0;
}

extern "C" int arl_rollout_begin(const arl_game* game, const arl_env_state* st, const arl_rollout* ro, void* stream) {
    int rc = check_env_args(game, st, ro);
    if (rc) return rc;
    ARL_REQUIRE(ro->observations, ARL_E_ARG, "null rollout arrays");
    const int64_t row = (int64_t)game->n_stack * OBS_FRAME;
    ARL_REQUIRE(arl::aligned16(ro->step_obs) && arl::aligned16(ro->observations), ARL_E_ALIGN, "16-byte alignment");
    hipLaunchKernelGGL(batch_begin_kernel, dim3((unsigned)st->n_env), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)ro->step_obs, (uint4*)ro->observations, (int)(row / 16), (int)ro->horizon, st->done_count);
    return arl::check_launch("batch_begin_kernel");
}

extern "C" int arl_env_reset(const arl_game* game, const arl_env_state* st, const arl_rollout* ro,
                             const uint8_t* flags_or_null, int32_t max_start_noops, void* stream) {
    int rc = check_env_args(game, st, ro);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL((frame_step_kernel<true>), dim3((unsigned)st->n_env), dim3(256), 0, s, *game,
                       *st, *ro, flags_or_null, 0, (int)max_start_noops);
    rc = arl::check_launch("frame_step_kernel<reset>");
    if (rc) return rc;
    hipLaunchKernelGGL(epoch_kernel, dim3(1), dim3(1), 0, s, st->epoch, st->next_reset ? st->launch_count : (int32_t*)nullptr);
    return arl::check_launch("epoch_kernel");
}

extern "C" int arl_preprocess_frames(const uint8_t* raw_a_or_null, const uint8_t* raw_b, int64_t n,
                                     int32_t resample_mode, uint8_t* out, void* stream) {
    ARL_REQUIRE(raw_b && out, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(n >= 0, ARL_E_ARG, "negative n");
    ARL_REQUIRE(resample_mode == ARL_RESAMPLE_BOX2X || resample_mode == ARL_RESAMPLE_NEAREST, ARL_E_ARG,
                "resample_mode must be ARL_RESAMPLE_BOX2X or ARL_RESAMPLE_NEAREST");
    ARL_REQUIRE(arl::aligned16(raw_b) && (!raw_a_or_null || arl::aligned16(raw_a_or_null)) &&
                    !(reinterpret_cast<uintptr_t>(out) & 7u), ARL_E_ALIGN, "frames must be 16-byte aligned");
    if (n == 0) return 0;
    hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream,
                       raw_a_or_null, raw_b, out, (int)resample_mode);
    return arl::check_launch("preprocess_kernel");
}
