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

#include "arl_common.h"

namespace {

constexpr int RAW_FRAME = ARL_RAW_H * ARL_RAW_W;          // 33600
constexpr int OBS_FRAME = ARL_OBS_H * ARL_OBS_W;          // 8320
constexpr int UNITS = OBS_FRAME / 8;                      // 1040 8-pixel output units
constexpr int UNITS_PER_ROW = ARL_OBS_W / 8;              // 10

enum : uint8_t { MODE_SKIP = 0, MODE_PUSH = 1, MODE_BLANK_PUSH = 2 };

struct Emu {
    int tick, lives;
    bool over;
};

// oracle/synth_ale.py: SynthALE.act
__device__ __forceinline__ int emu_act(Emu& e, int code, int start_lives, int life_period) {
    if (e.over) return 0;
    e.tick += 1;
    int r = 0;
    if ((7 * e.tick + 3 * code) % 97 == 0) r += 1 + e.tick % 3;
    if ((5 * e.tick + code) % 89 == 0) r -= 1;
    if (start_lives > 0) {
        if (e.tick % life_period == 0) {
            e.lives -= 1;
            if (e.lives == 0) e.over = true;
        }
    } else if (e.tick >= life_period * 5) {
        e.over = true;
    }
    return r;
}

// atari_env.py:172-179 (_life_reset): NOOP, FIRE if present, UP if present
__device__ __forceinline__ void press_start(Emu& e, const arl_game& g, bool has_fire, bool has_up,
                                            int& env_lives) {
    emu_act(e, 0, g.start_lives, g.life_period);
    if (has_fire) emu_act(e, 1, g.start_lives, g.life_period);
    if (has_up) emu_act(e, 2, g.start_lives, g.life_period);
    env_lives = e.lives;
}

__device__ __forceinline__ bool has_code(const arl_game& g, int code) {
    bool f = false;
    for (int i = 0; i < g.n_actions; ++i) f |= (g.action_set[i] == code);
    return f;
}

__global__ __launch_bounds__(256) void act_step_kernel(
    const arl_game g, const arl_env_state st, const arl_rollout ro, const float* __restrict__ prob,
    const float* __restrict__ value, const double* __restrict__ uniforms,
    const uint8_t* __restrict__ active, int step, int mid_batch_reset, double max_path_length,
    double discount) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= st.n_env) return;
    if (active && !active[e]) {
        st.reset_flag[e] = 0;
        st.frame_mode[e] = MODE_SKIP;
        return;
    }
    const int T = ro.horizon, A = g.n_actions;
    const int64_t row = e * T + step;

    // ---- action: weighted_sample_n (special.py:22-27) + scatter (sampler.py:143-145)
    const float* p = prob + e * A;
    const double u = uniforms[e];
    float c = 0.f;
    int k = 0;
    for (int j = 0; j < A; ++j) {
        const float pj = p[j];
        ro.prob[row * A + j] = pj;
        c += pj;
        k += ((double)c < u) ? 1 : 0;
    }
    const int a_idx = k < A - 1 ? k : A - 1;
    ro.actions[row] = (uint8_t)a_idx;
    ro.value[row] = value[e];

    st.reset_flag[e] = 0;
    if (!mid_batch_reset && st.frozen[e]) {              // worker.py:80 (env sits out the batch)
        st.frame_mode[e] = MODE_SKIP;
        return;
    }

    // ---- AtariEnv.step (atari_env.py:65-78)
    const bool has_fire = has_code(g, 1), has_up = has_code(g, 2);
    Emu emu = {st.tick[e], st.emu_lives[e], st.over[e] != 0};
    int env_lives = st.env_lives[e];
    const int phase = st.phase[e];
    const int code = g.action_set[a_idx];
    float reward = 0.f;
    for (int i = 0; i < g.frame_skip - 1; ++i) reward += (float)emu_act(emu, code, g.start_lives, g.life_period);
    int fa = (phase + emu.tick) % g.n_frames;            // _get_screen(1)
    reward += (float)emu_act(emu, code, g.start_lives, g.life_period);
    int fb = (phase + emu.tick) % g.n_frames;            // _update_obs: _get_screen(2)
    uint8_t mode = MODE_PUSH;
    const float raw = reward;
    if (g.clip_reward) reward = (reward > 0.f) ? 1.f : ((reward < 0.f) ? -1.f : 0.f);

    bool need_reset = emu.over;                          // atari_env.py:186
    const bool lost = (emu.lives < env_lives) && (emu.lives > 0);   // :167
    bool done;
    if (g.episodic_lives) {
        if (lost) {                                      // :188-190
            press_start(emu, g, has_fire, has_up, env_lives);
            fa = -1;
            fb = (phase + emu.tick) % g.n_frames;
            mode = MODE_BLANK_PUSH;
        }
        done = lost || need_reset;
    } else {
        if (lost) press_start(emu, g, has_fire, has_up, env_lives);   // :181-183
        done = emu.over;
    }

    // ---- TrajInfo.step (sampler/util.py:92-101)
    int t_len = st.traj_len[e] + 1;
    float t_ret = st.traj_ret[e] + reward;
    float t_raw = st.traj_raw[e] + (g.clip_reward ? raw : reward);
    int t_nz = st.traj_nonzero[e] + (reward != 0.f ? 1 : 0);
    double cur = st.traj_curdisc[e];
    float t_disc = st.traj_disc[e] + (float)cur * reward;
    cur *= discount;

    // ---- collector rules (worker.py:42-50 / :84-95)
    const bool over_len = (double)t_len > max_path_length;
    const bool reset_cond = g.episodic_lives ? need_reset : true;    // info.get("need_reset", True)
    const bool hit = over_len || (done && reset_cond);
    if (hit) {
        done = true;
        if (over_len && g.episodic_lives) need_reset = true;
        const int slot = atomicAdd(st.done_count, 1);
        if (slot < st.done_capacity) {
            st.done_int[slot * 3] = (int)e;
            st.done_int[slot * 3 + 1] = t_len;
            st.done_int[slot * 3 + 2] = t_nz;
            st.done_flt[slot * 3] = t_ret;
            st.done_flt[slot * 3 + 1] = t_raw;
            st.done_flt[slot * 3 + 2] = t_disc;
        }
        t_len = 0; t_ret = 0.f; t_raw = 0.f; t_nz = 0; t_disc = 0.f; cur = 1.0;
        if (mid_batch_reset) {
            st.reset_flag[e] = 1;                        // env.reset() happens in frame_step
        } else {
            st.frozen[e] = 1;                            // worker.py:89
            mode = MODE_SKIP;                            // obs not written (worker.py:96-99)
        }
    }

    ro.rewards[row] = reward;
    ro.dones[row] = done ? 1 : 0;
    if (ro.raw_reward) ro.raw_reward[row] = raw;
    if (ro.need_reset) ro.need_reset[row] = need_reset ? 1 : 0;

    st.tick[e] = emu.tick; st.emu_lives[e] = emu.lives; st.over[e] = emu.over ? 1 : 0;
    st.env_lives[e] = env_lives;
    st.traj_len[e] = t_len; st.traj_ret[e] = t_ret; st.traj_raw[e] = t_raw;
    st.traj_nonzero[e] = t_nz; st.traj_disc[e] = t_disc; st.traj_curdisc[e] = cur;
    st.frame_a[e] = fa; st.frame_b[e] = fb; st.frame_mode[e] = mode;
}

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

// packed u16x2 max (v_pk_max_u16)
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
    const us2 r = __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b));
    return __builtin_bit_cast(uint32_t, r);
}

// One 32-bit word = 4 input pixels of two rows (top, bot) of two frames (a, b):
// max over frames, 2x2 rounded box mean -> 2 output pixels in bits [0,8) and [8,16).
// Bytes are split into even/odd u16 lanes so every step is a packed-16 op and no
// intermediate can overflow (4 * 255 + 2 < 65536).
__device__ __forceinline__ uint32_t box_word(uint32_t a_top, uint32_t b_top, uint32_t a_bot, uint32_t b_bot) {
    const uint32_t M = 0x00ff00ffu;
    const uint32_t te = pk_max(a_top & M, b_top & M), to = pk_max((a_top >> 8) & M, (b_top >> 8) & M);
    const uint32_t be = pk_max(a_bot & M, b_bot & M), bo = pk_max((a_bot >> 8) & M, (b_bot >> 8) & M);
    const uint32_t sum = te + to + be + bo + 0x00020002u;      // lanes: px(0,1) and px(2,3) column pairs
    const uint32_t avg = (sum >> 2) & M;
    return (avg & 0xffu) | ((avg >> 8) & 0xff00u);
}

// max / crop / 2x2 rounded box for one 8-pixel output unit (atari_env.py:154-155).
// a0,a1: rows 2y,2y+1 of frame A (16 bytes each); b0,b1 same for frame B.
__device__ __forceinline__ uint2 box8(uint4 a0, uint4 a1, uint4 b0, uint4 b1) {
    const uint32_t lo = box_word(a0.x, b0.x, a1.x, b1.x) | (box_word(a0.y, b0.y, a1.y, b1.y) << 16);
    const uint32_t hi = box_word(a0.z, b0.z, a1.z, b1.z) | (box_word(a0.w, b0.w, a1.w, b1.w) << 16);
    return make_uint2(lo, hi);
}

// Resolve a pending reset for env e (lane 0 of its workgroup).  Order of the
// start-noop draws inside a stream = env order among the envs that reset in
// this launch (the reference worker loops over its envs, worker.py:38-50).
__device__ void resolve_reset(const arl_game& g, const arl_env_state& st, int64_t e,
                              const uint8_t* flags, int max_start_noops, int parity) {
    const int64_t per = st.envs_per_stream;
    const int64_t w = e / per, g0 = w * per;
    int rank = 0;
    for (int64_t i = g0; i < e; ++i) rank += (flags ? (flags[i] != 0) : 1);
    const int64_t n_streams = (st.n_env + per - 1) / per;
    const int64_t cur = st.noop_cursor[parity * n_streams + w];
    int noops = 0;
    if (max_start_noops > 0)                              // randint(0, 1) draws nothing
        noops = st.noop_ring[w * st.noop_ring_len + (cur + rank) % st.noop_ring_len];
    Emu emu = {0, g.start_lives, false};                  // ale.reset_game(), atari_env.py:94
    int env_lives = 0;
    press_start(emu, g, has_code(g, 1), has_code(g, 2), env_lives);   // :96
    for (int i = 0; i < noops; ++i) emu_act(emu, 0, g.start_lives, g.life_period);   // :97-98
    st.tick[e] = emu.tick; st.emu_lives[e] = emu.lives; st.over[e] = emu.over ? 1 : 0;
    st.env_lives[e] = env_lives;
    st.frame_a[e] = -1;
    st.frame_b[e] = (st.phase[e] + emu.tick) % g.n_frames;
    st.frame_mode[e] = MODE_BLANK_PUSH;                   // _reset_obs + one _update_obs (:95,99)
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
    }
    __syncthreads();
    const int mode = s_mode;
    if (mode == MODE_SKIP) return;

    const int F = g.n_stack;
    const uint8_t* fb = g.bank + (int64_t)s_fb * RAW_FRAME;
    const uint8_t* fa = (s_fa >= 0) ? g.bank + (int64_t)s_fa * RAW_FRAME : nullptr;
    uint8_t* cur = ro.step_obs + e * (int64_t)F * OBS_FRAME;
    uint8_t* dst = nullptr;
    if (!RESET_ONLY && step + 1 < ro.horizon)             // worker.py:52-53
        dst = ro.observations + (e * ro.horizon + step + 1) * (int64_t)F * OBS_FRAME;

    for (int un = tid; un < UNITS; un += blockDim.x) {
        const int y = un / UNITS_PER_ROW, xb = un - y * UNITS_PER_ROW;
        const int src = (2 * y) * ARL_RAW_W + xb * 16;
        const uint4 b0 = *reinterpret_cast<const uint4*>(fb + src);
        const uint4 b1 = *reinterpret_cast<const uint4*>(fb + src + ARL_RAW_W);
        uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
        if (fa) {
            a0 = *reinterpret_cast<const uint4*>(fa + src);
            a1 = *reinterpret_cast<const uint4*>(fa + src + ARL_RAW_W);
        }
        const uint2 img = box8(a0, a1, b0, b1);
        const int o = un * 8;
        // stack: oldest -> newest (atari_env.py:156-157)
        for (int f = 0; f < F - 1; ++f) {
            uint2 prev = make_uint2(0, 0);
            if (mode == MODE_PUSH) prev = *reinterpret_cast<const uint2*>(cur + (f + 1) * OBS_FRAME + o);
            *reinterpret_cast<uint2*>(cur + f * OBS_FRAME + o) = prev;
            if (dst) *reinterpret_cast<uint2*>(dst + f * OBS_FRAME + o) = prev;
        }
        *reinterpret_cast<uint2*>(cur + (F - 1) * OBS_FRAME + o) = img;
        if (dst) *reinterpret_cast<uint2*>(dst + (F - 1) * OBS_FRAME + o) = img;
    }
}

// bump the launch epoch AFTER a frame_step launch (single lane)
__global__ void epoch_kernel(int32_t* epoch) { epoch[0] += 1; }

__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* __restrict__ a,
                                                         const uint8_t* __restrict__ b,
                                                         uint8_t* __restrict__ out) {
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
        *reinterpret_cast<uint2*>(out + i * OBS_FRAME + un * 8) = box8(a0, a1, b0, b1);
    }
}

int check_env_args(const arl_game* g, const arl_env_state* st, const arl_rollout* ro) {
    if (!g || !st || !ro) { arl::set_error("env: null struct"); return ARL_E_ARG; }
    if (!g->bank || g->n_frames <= 0 || g->n_actions <= 0 || g->n_actions > ARL_MAX_ACTIONS ||
        g->frame_skip < 1 || g->n_stack < 1 || g->life_period <= 0) {
        arl::set_error("env: bad game description"); return ARL_E_ARG;
    }
    if (st->n_env <= 0 || st->envs_per_stream <= 0 || st->noop_ring_len <= 0 || !st->tick ||
        !st->noop_cursor || !st->epoch || !st->noop_ring || !st->done_count) {
        arl::set_error("env: bad state description"); return ARL_E_ARG;
    }
    if (!ro->step_obs || ro->horizon <= 0) { arl::set_error("env: bad rollout description"); return ARL_E_ARG; }
    if (!arl::aligned16(g->bank) || (reinterpret_cast<uintptr_t>(ro->step_obs) & 7u) ||
        (ro->observations && (reinterpret_cast<uintptr_t>(ro->observations) & 7u))) {
        arl::set_error("env: bank must be 16-byte, observations 8-byte aligned"); return ARL_E_ALIGN;
    }
    return 0;
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
    hipLaunchKernelGGL(epoch_kernel, dim3(1), dim3(1), 0, s, st->epoch);
    return arl::check_launch("epoch_kernel");
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
    hipLaunchKernelGGL(epoch_kernel, dim3(1), dim3(1), 0, s, st->epoch);
    return arl::check_launch("epoch_kernel");
}

extern "C" int arl_preprocess_frames(const uint8_t* raw_a_or_null, const uint8_t* raw_b, int64_t n,
                                     uint8_t* out, void* stream) {
    ARL_REQUIRE(raw_b && out, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(n >= 0, ARL_E_ARG, "negative n");
    ARL_REQUIRE(arl::aligned16(raw_b) && (!raw_a_or_null || arl::aligned16(raw_a_or_null)) &&
                    !(reinterpret_cast<uintptr_t>(out) & 7u), ARL_E_ALIGN, "frames must be 16-byte aligned");
    if (n == 0) return 0;
    hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream,
                       raw_a_or_null, raw_b, out);
    return arl::check_launch("preprocess_kernel");
}
