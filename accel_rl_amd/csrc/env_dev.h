// Device-side pieces of the env step shared by env.hip (arl_env_step and its two-launch form) and serve_step.hip
// (arl_env_step_served): the synthetic emulator, AtariEnv / collector / TrajInfo rules as pure functions on registers,
// the pixel pipeline (max / crop / 2x2 box / stack) and the frame push.  Reference lines: see env.hip.
#pragma once
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

// One env's mutable state in registers.
struct EnvRegs {
    int tick, emu_lives, env_lives, phase, traj_len, traj_nz, over, frozen;
    float traj_ret, traj_raw, traj_disc;
    double curdisc;
};

__device__ __forceinline__ EnvRegs load_env(const arl_env_state& st, const int64_t e) {
    EnvRegs s;
    s.tick = st.tick[e]; s.emu_lives = st.emu_lives[e]; s.env_lives = st.env_lives[e]; s.phase = st.phase[e];
    s.traj_len = st.traj_len[e]; s.traj_nz = st.traj_nonzero[e]; s.over = st.over[e]; s.frozen = st.frozen[e];
    s.traj_ret = st.traj_ret[e]; s.traj_raw = st.traj_raw[e]; s.traj_disc = st.traj_disc[e];
    s.curdisc = st.traj_curdisc[e];
    return s;
}

// Everything one agent step of one env decides, as values: the scalar part of the step is load_env ->
// step_compute (pure) -> step_commit (stores).  The fused step kernel runs the first two on EVERY lane of the
// env's workgroup (same addresses: the loads broadcast), so all lanes know which frames to push without
// waiting for the one lane that commits.
struct StepOut {
    int stepped, emulated;          // 0: env inactive this launch / env frozen for the rest of the batch
    int a_idx;
    float reward, raw;
    int done, need_reset, hit, reset_flag, set_frozen;
    int fa, fb, mode;
    EnvRegs s;                      // state after the step (a flagged reset not yet applied)
    int rec_len, rec_nz;            // completed-trajectory record (hit only)
    float rec_ret, rec_raw, rec_disc;
};

// weighted_sample_n (special.py:22-27): k = #{j : cumsum_j(p) < u}, clamped; fp32 sequential cumsum, f64 compare
__device__ __forceinline__ int sample_action(const float* __restrict__ p, const int A, const double u) {
    float c = 0.f;
    int k = 0;
    for (int j = 0; j < A; ++j) {
        c += p[j];
        k += ((double)c < u) ? 1 : 0;
    }
    return k < A - 1 ? k : A - 1;
}

__device__ __forceinline__ StepOut step_compute(const arl_game& g, const EnvRegs& in, const int a_idx,
                                                const bool is_active, const int mid_batch_reset,
                                                const double max_path_length, const double discount) {
    StepOut o;
    o.s = in;
    o.stepped = is_active; o.emulated = 0; o.a_idx = 0; o.reward = 0.f; o.raw = 0.f;
    o.done = 0; o.need_reset = 0; o.hit = 0; o.reset_flag = 0; o.set_frozen = 0;
    o.fa = -1; o.fb = 0; o.mode = MODE_SKIP;
    o.rec_len = 0; o.rec_nz = 0; o.rec_ret = 0.f; o.rec_raw = 0.f; o.rec_disc = 0.f;
    if (!is_active) return o;
    o.a_idx = a_idx;
    if (!mid_batch_reset && in.frozen) return o;         // worker.py:80 (env sits out the batch)
    o.emulated = 1;

    // ---- AtariEnv.step (atari_env.py:65-78)
    const bool has_fire = has_code(g, 1), has_up = has_code(g, 2);
    Emu emu = {in.tick, in.emu_lives, in.over != 0};
    int env_lives = in.env_lives;
    const int phase = in.phase;
    const int code = g.action_set[a_idx];
    float reward = 0.f;
    for (int i = 0; i < g.frame_skip - 1; ++i) reward += (float)emu_act(emu, code, g.start_lives, g.life_period);
    int fa = (phase + emu.tick) % g.n_frames;            // _get_screen(1)
    reward += (float)emu_act(emu, code, g.start_lives, g.life_period);
    int fb = (phase + emu.tick) % g.n_frames;            // _update_obs: _get_screen(2)
    int mode = MODE_PUSH;
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
    int t_len = in.traj_len + 1;
    float t_ret = in.traj_ret + reward;
    float t_raw = in.traj_raw + (g.clip_reward ? raw : reward);
    int t_nz = in.traj_nz + (reward != 0.f ? 1 : 0);
    double cur = in.curdisc;
    float t_disc = in.traj_disc + (float)cur * reward;
    cur *= discount;

    // ---- collector rules (worker.py:42-50 / :84-95)
    const bool over_len = (double)t_len > max_path_length;
    const bool reset_cond = g.episodic_lives ? need_reset : true;    // info.get("need_reset", True)
    const bool hit = over_len || (done && reset_cond);
    if (hit) {
        done = true;
        if (over_len && g.episodic_lives) need_reset = true;
        o.rec_len = t_len; o.rec_nz = t_nz; o.rec_ret = t_ret; o.rec_raw = t_raw; o.rec_disc = t_disc;
        t_len = 0; t_ret = 0.f; t_raw = 0.f; t_nz = 0; t_disc = 0.f; cur = 1.0;
        if (mid_batch_reset) {
            o.reset_flag = 1;                            // env.reset() follows (resolve_reset / reset_regs)
        } else {
            o.set_frozen = 1;                            // worker.py:89
            o.s.frozen = 1;
            mode = MODE_SKIP;                            // obs not written (worker.py:96-99)
        }
    }
    o.reward = reward; o.raw = raw; o.done = done; o.need_reset = need_reset; o.hit = hit;
    o.fa = fa; o.fb = fb; o.mode = mode;
    o.s.tick = emu.tick; o.s.emu_lives = emu.lives; o.s.over = emu.over ? 1 : 0; o.s.env_lives = env_lives;
    o.s.traj_len = t_len; o.s.traj_ret = t_ret; o.s.traj_raw = t_raw; o.s.traj_nz = t_nz; o.s.traj_disc = t_disc;
    o.s.curdisc = cur;
    return o;
}

// The stores of one step (one lane): scatter of the served action (sampler.py:143-145), rollout row, env state,
// completed-trajectory record.  apply_frames: also the frame hand-off fields (after a resolved reset they
// already hold the reset's frames).
template <bool COPY_PROB = true>
__device__ __forceinline__ void step_commit(const arl_game& g, const arl_env_state& st, const arl_rollout& ro,
                                            const float* __restrict__ p, const float v, const StepOut& o,
                                            const int64_t e, const int step) {
    st.reset_flag[e] = (uint8_t)o.reset_flag;
    if (!o.stepped) { st.frame_mode[e] = MODE_SKIP; return; }
    const int A = g.n_actions;
    const int64_t row = e * ro.horizon + step;
    if (COPY_PROB)
        for (int j = 0; j < A; ++j) ro.prob[row * A + j] = p[j];
    ro.actions[row] = (uint8_t)o.a_idx;
    ro.value[row] = v;
    if (!o.emulated) { st.frame_mode[e] = MODE_SKIP; return; }
    if (o.hit) {
        const int slot = atomicAdd(st.done_count, 1);
        if (slot < st.done_capacity) {
            st.done_int[slot * 3] = (int)e;
            st.done_int[slot * 3 + 1] = o.rec_len;
            st.done_int[slot * 3 + 2] = o.rec_nz;
            st.done_flt[slot * 3] = o.rec_ret;
            st.done_flt[slot * 3 + 1] = o.rec_raw;
            st.done_flt[slot * 3 + 2] = o.rec_disc;
        }
        if (o.set_frozen) st.frozen[e] = 1;
    }
    ro.rewards[row] = o.reward;
    ro.dones[row] = o.done ? 1 : 0;
    if (ro.raw_reward) ro.raw_reward[row] = o.raw;
    if (ro.need_reset) ro.need_reset[row] = o.need_reset ? 1 : 0;
    st.tick[e] = o.s.tick; st.emu_lives[e] = o.s.emu_lives; st.over[e] = (uint8_t)o.s.over;
    st.env_lives[e] = o.s.env_lives;
    st.traj_len[e] = o.s.traj_len; st.traj_ret[e] = o.s.traj_ret; st.traj_raw[e] = o.s.traj_raw;
    st.traj_nonzero[e] = o.s.traj_nz; st.traj_disc[e] = o.s.traj_disc; st.traj_curdisc[e] = o.s.curdisc;
    st.frame_a[e] = o.fa; st.frame_b[e] = o.fb; st.frame_mode[e] = (uint8_t)o.mode;
}

// AtariEnv.reset on registers (atari_env.py:93-100): reset_game, press start, `noops` no-op frames; the new
// observation is a blank stack with one fresh frame on top.
__device__ __forceinline__ void reset_regs(const arl_game& g, EnvRegs& s, const int noops, int& fa, int& fb, int& mode) {
    Emu emu = {0, g.start_lives, false};                  // ale.reset_game(), atari_env.py:94
    int env_lives = 0;
    press_start(emu, g, has_code(g, 1), has_code(g, 2), env_lives);   // :96
    for (int i = 0; i < noops; ++i) emu_act(emu, 0, g.start_lives, g.life_period);   // :97-98
    s.tick = emu.tick; s.emu_lives = emu.lives; s.over = emu.over ? 1 : 0; s.env_lives = env_lives;
    fa = -1;
    fb = (s.phase + emu.tick) % g.n_frames;
    mode = MODE_BLANK_PUSH;                               // _reset_obs + one _update_obs (:95,99)
}

// Will step_compute flag this env for a mid-batch reset in its NEXT step?  None of its rules depends on the
// sampled action (the action only enters the reward): game over and life loss follow from the emulator's tick,
// over-length from the trajectory length.  Every env's own workgroup evaluates this once its state is final
// and leaves it in st.next_reset for the next launch (ping-pong by launch parity, like the cursors), where
// the other workgroups of the worker stream read it: each then knows its env's rank among the stream's resets
// (= which start no-op draw is its own) and the stream leader their number, without a second launch.
__device__ __forceinline__ bool will_reset(const arl_game& g, const EnvRegs& s, const double max_path_length) {
    Emu emu = {s.tick, s.emu_lives, s.over != 0};
    for (int k = 0; k < g.frame_skip; ++k) emu_act(emu, 0, g.start_lives, g.life_period);
    const bool need_reset = emu.over;
    const bool lost = (emu.lives < s.env_lives) && (emu.lives > 0);
    const bool done = g.episodic_lives ? (lost || need_reset) : emu.over;
    const bool over_len = (double)(s.traj_len + 1) > max_path_length;
    const bool reset_cond = g.episodic_lives ? need_reset : true;
    return over_len || (done && reset_cond);
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

// cv2.INTER_NEAREST for the exact 2x decimation (what atari_env.py:155 names but does not get, SURVEY a-11):
// dst(y, x) = src(2y, 2x) -- OpenCV's nearest-neighbour source index is floor(dst * scale), no half-pixel shift.
// One 32-bit word = 4 input pixels of row 2y of frames a, b -> the even bytes of their max, packed as in box_word.
__device__ __forceinline__ uint32_t nearest_word(uint32_t a_top, uint32_t b_top) {
    const uint32_t m = pk_max(a_top & 0x00ff00ffu, b_top & 0x00ff00ffu);
    return (m & 0xffu) | ((m >> 8) & 0xff00u);
}

// One 8-pixel output unit under the game's resample mode (wave-uniform switch).
__device__ __forceinline__ uint2 resample8(const int mode, uint4 a0, uint4 a1, uint4 b0, uint4 b1) {
    if (mode == ARL_RESAMPLE_NEAREST) {
        const uint32_t lo = nearest_word(a0.x, b0.x) | (nearest_word(a0.y, b0.y) << 16);
        const uint32_t hi = nearest_word(a0.z, b0.z) | (nearest_word(a0.w, b0.w) << 16);
        return make_uint2(lo, hi);
    }
    return box8(a0, a1, b0, b1);
}

// Resolve a pending reset for env e (lane 0 of its workgroup).  Order of the
// start-noop draws inside a stream = env order among the envs that reset in
// this launch (the reference worker loops over its envs, worker.py:38-50).
__device__ void resolve_reset(const arl_game& g, const arl_env_state& st, int64_t e,
                              const uint8_t* flags, int max_start_noops, int parity, int known_rank = -1) {
    const int64_t per = st.envs_per_stream;
    const int64_t w = e / per, g0 = w * per;
    int rank = known_rank < 0 ? 0 : known_rank;
    if (known_rank < 0)
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

// The pixel part of one env step, one workgroup: new frame = rounded 2x2 box of the cropped max of bank
// frames fa (< 0: none) and fb; new stack = the previous one (`prev`; zeros in MODE_BLANK_PUSH) shifted by one
// with the new frame on top, written to out0 and (if given) out1.  prev may alias out0: a thread reads and
// writes the same pixels of every plane, oldest plane first.
// The frame push in two halves, so that a caller can put other work under the loads' latency: FramePush::load
// issues EVERY load of the thread's units (5 per thread, the last one for 16 threads only: 4 raw 16-byte rows +
// the previous stack's 8-byte pieces each) before anything consumes one; FramePush::store boxes and writes.
constexpr int MAX_STACK = 4;                              // previous planes kept in registers (n_stack <= 4 fast path)
// The older planes of the stack (planes 1 .. F-1 of the previous observation become planes 0 .. F-2 of the new one) move
// 16 bytes per lane: a wave's store then covers whole 128-byte lines (as 8-byte pieces per pixel unit the same bytes
// left the L2 as 64-byte write requests, and this copy was 190 of the kernel's 355 us at 16 384 envs:
// tools/env_step_bound.sh).  A thread owns a 16-byte COLUMN through all planes -- it reads piece q of planes 1 .. F-1
// and writes piece q of planes 0 .. F-2.  WIDE needs the new observation to live elsewhere than the previous one
// (single_write: rows t and t + 1 of the rollout buffer): where step_obs is overwritten in place (prev == out0), the
// thread that writes the new frame's pixels must be the one that read them from the newest old plane, and those are
// 8-byte units -- the narrow copy (!WIDE: per pixel unit, 8 bytes per plane, all reads of a thread before its writes).
constexpr int COPY_COLS = OBS_FRAME / 16;                 // 520 pieces per plane

// NT = threads of the workgroup (256: 5 pixel units and 3 copy pieces per thread, the last ones for 16 / 8 threads only;
// 1024, serve_step.hip: 2 and 1)
template <bool WIDE, int NT = 256>
struct FramePush {
    static constexpr int PUSH_ITERS = (UNITS + NT - 1) / NT;
    static constexpr int COPY_ITERS = (COPY_COLS + NT - 1) / NT;
    uint4 a0[PUSH_ITERS], a1[PUSH_ITERS], b0[PUSH_ITERS], b1[PUSH_ITERS];
    uint4 cp[WIDE ? COPY_ITERS : 1][MAX_STACK - 1];
    uint2 pv[WIDE ? 1 : PUSH_ITERS][MAX_STACK - 1];

    template <int KO = 0>
    __device__ __forceinline__ void load(const arl_game& g, const int fa_i, const int fb_i, const int mode,
                                         const uint8_t* prev, const int tid) {
        const int F = g.n_stack;
        const uint8_t* fb = g.bank + (int64_t)fb_i * RAW_FRAME;
        const uint8_t* fa = (fa_i >= 0) ? g.bank + (int64_t)fa_i * RAW_FRAME : nullptr;
#pragma unroll
        for (int it = 0; it < PUSH_ITERS; ++it) {
            const int un = tid + it * NT;
            a0[it] = make_uint4(0, 0, 0, 0); a1[it] = a0[it]; b0[it] = a0[it]; b1[it] = a0[it];
            if (un < UNITS) {
                const int y = un / UNITS_PER_ROW, xb = un - y * UNITS_PER_ROW;
                const int src = (2 * y) * ARL_RAW_W + xb * 16;
                b0[it] = *reinterpret_cast<const uint4*>(fb + src);
                b1[it] = *reinterpret_cast<const uint4*>(fb + src + ARL_RAW_W);
                if (fa) {
                    a0[it] = *reinterpret_cast<const uint4*>(fa + src);
                    a1[it] = *reinterpret_cast<const uint4*>(fa + src + ARL_RAW_W);
                }
            }
            if (!WIDE) {
#pragma unroll
                for (int f = 0; f < MAX_STACK - 1; ++f) {
                    pv[it][f] = make_uint2(0, 0);                // MODE_BLANK_PUSH: a blank history
                    if (mode == MODE_PUSH && un < UNITS && f < F - 1)
                        pv[it][f] = *reinterpret_cast<const uint2*>(prev + (f + 1) * OBS_FRAME + un * 8);
                }
            }
        }
        if (WIDE) {
#pragma unroll
            for (int it = 0; it < COPY_ITERS; ++it) {
                const int q = tid + it * NT;
#pragma unroll
                for (int f = 0; f < MAX_STACK - 1; ++f) {
                    cp[it][f] = make_uint4(0, 0, 0, 0);
                    if (mode == MODE_PUSH && KO != 3 && q < COPY_COLS && f < F - 1)
                        cp[it][f] = reinterpret_cast<const uint4*>(prev + (f + 1) * OBS_FRAME)[q];
                }
            }
        }
    }

    // stack: oldest -> newest (atari_env.py:156-157)
    // lds_img (WIDE only): the new stacked observation is ALSO left in LDS, [F][OBS_FRAME] bytes (16-byte aligned), for a
    // consumer inside the same launch (serve_step.hip: the first convolution)
    template <int KO = 0>
    __device__ __forceinline__ void store(const arl_game& g, uint8_t* out0, uint8_t* out1, const int tid,
                                          uint8_t* lds_img = nullptr) const {
        const int F = g.n_stack;
#pragma unroll
        for (int it = 0; it < PUSH_ITERS; ++it) {
            const int un = tid + it * NT;
            if (un >= UNITS) continue;
            const uint2 img = resample8(g.resample_mode, a0[it], a1[it], b0[it], b1[it]);
            const int o = un * 8;
            if (!WIDE) {
#pragma unroll
                for (int f = 0; f < MAX_STACK - 1; ++f)
                    if (f < F - 1) {
                        *reinterpret_cast<uint2*>(out0 + f * OBS_FRAME + o) = pv[it][f];
                        if (out1) *reinterpret_cast<uint2*>(out1 + f * OBS_FRAME + o) = pv[it][f];
                    }
            }
            *reinterpret_cast<uint2*>(out0 + (F - 1) * OBS_FRAME + o) = img;
            if (out1) *reinterpret_cast<uint2*>(out1 + (F - 1) * OBS_FRAME + o) = img;
            if (WIDE && lds_img) *reinterpret_cast<uint2*>(lds_img + (F - 1) * OBS_FRAME + o) = img;
        }
        if (!WIDE || KO == 2 || KO == 3) return;
#pragma unroll
        for (int it = 0; it < COPY_ITERS; ++it) {
            const int q = tid + it * NT;
            if (q >= COPY_COLS) continue;
#pragma unroll
            for (int f = 0; f < MAX_STACK - 1; ++f)
                if (f < F - 1) {
                    reinterpret_cast<uint4*>(out0 + f * OBS_FRAME)[q] = cp[it][f];
                    if (out1) reinterpret_cast<uint4*>(out1 + f * OBS_FRAME)[q] = cp[it][f];
                    if (lds_img) reinterpret_cast<uint4*>(lds_img + f * OBS_FRAME)[q] = cp[it][f];
                }
        }
    }
};

// The pixel part of one env step, one workgroup: new frame = rounded 2x2 box of the cropped max of bank
// frames fa (< 0: none) and fb; new stack = the previous one (`prev`; zeros in MODE_BLANK_PUSH) shifted by one
// with the new frame on top, written to out0 and (if given) out1.  prev may alias out0: a thread reads and
// writes the same pixels of every plane, oldest plane first.  (Any stack depth; the step kernel's fast path
// is FramePush.)
__device__ __forceinline__ void push_frame(const arl_game& g, const int fa_i, const int fb_i, const int mode,
                                           const uint8_t* prev, uint8_t* out0, uint8_t* out1, const int tid) {
    const int F = g.n_stack;
    const uint8_t* fb = g.bank + (int64_t)fb_i * RAW_FRAME;
    const uint8_t* fa = (fa_i >= 0) ? g.bank + (int64_t)fa_i * RAW_FRAME : nullptr;
    for (int un = tid; un < UNITS; un += 256) {
        const int y = un / UNITS_PER_ROW, xb = un - y * UNITS_PER_ROW;
        const int src = (2 * y) * ARL_RAW_W + xb * 16;
        const uint4 b0 = *reinterpret_cast<const uint4*>(fb + src);
        const uint4 b1 = *reinterpret_cast<const uint4*>(fb + src + ARL_RAW_W);
        uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
        if (fa) {
            a0 = *reinterpret_cast<const uint4*>(fa + src);
            a1 = *reinterpret_cast<const uint4*>(fa + src + ARL_RAW_W);
        }
        const uint2 img = resample8(g.resample_mode, a0, a1, b0, b1);
        const int o = un * 8;
        for (int f = 0; f < F - 1; ++f) {
            uint2 pv = make_uint2(0, 0);
            if (mode == MODE_PUSH) pv = *reinterpret_cast<const uint2*>(prev + (f + 1) * OBS_FRAME + o);
            *reinterpret_cast<uint2*>(out0 + f * OBS_FRAME + o) = pv;
            if (out1) *reinterpret_cast<uint2*>(out1 + f * OBS_FRAME + o) = pv;
        }
        *reinterpret_cast<uint2*>(out0 + (F - 1) * OBS_FRAME + o) = img;
        if (out1) *reinterpret_cast<uint2*>(out1 + (F - 1) * OBS_FRAME + o) = img;
    }
}

// host: the argument checks every env entry point starts with
inline int check_env_args(const arl_game* g, const arl_env_state* st, const arl_rollout* ro) {
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
