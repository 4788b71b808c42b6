/*
 * accel_rl_hip.h -- C-ABI of libaccel_rl_hip.so (hand-written HIP for gfx950 / MI355X).
 *
 * The reference (astooke/accel_rl) is pure Python and has NO FFI boundary of its
 * own; its hot path is Python/numpy loops and Theano graphs.  The
 * entry points below are what a binding for that path would bind: one function
 * per reference routine on the path, each citing the reference code it replaces
 * (paths relative to the reference root).  INTEGRATION.md shows the ctypes stub
 * a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (no allocation
 *     inside, no torch types); sizes are plain integers;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every
 *     call only ENQUEUES work on it and is hipGraph-capturable;
 *   - return value: 0 = ok; < 0 = argument error (ARL_E_*); > 0 = hipError_t of
 *     the failed launch.  arl_last_error() returns a thread-local message;
 *   - batch arrays are "env-major": flat index = env * horizon + t
 *     (accel_rl/buffers/batch.py:59-76);
 *   - results: integer / byte / index outputs are bit-exact with the reference;
 *     floating point follows the reference's own dtype walk (see `promo`).
 */
#ifndef ACCEL_RL_HIP_H
#define ACCEL_RL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARL_ABI_VERSION 4

#define ARL_E_ARG      (-1)   /* null pointer / non-positive size                 */
#define ARL_E_RANGE    (-2)   /* size outside what the kernels support             */
#define ARL_E_ALIGN    (-3)   /* pointer not aligned as documented                 */

/* numpy promotion the reference arithmetic is reproduced under:
 * NEP50  = numpy >= 2 (python float x float32 -> float32), bit-exact with the
 *          reference as it runs today;
 * LEGACY = numpy 1.x (python float x float32 scalar -> float64), as in 2018;
 * ASSOC  = (scans only) LEGACY's operand types with the recurrence evaluated as a wavefront suffix scan of affine
 *          maps instead of a sequential walk: f64 sums are reassociated, results agree with both exact modes to
 *          1e-5 (the tolerance BASELINE.json states for returns / advantages) but not bit for bit.  Used for
 *          horizons of 96 .. 512 steps, where it is the faster kernel; others take the LEGACY walk.            */
#define ARL_PROMO_NEP50   0
#define ARL_PROMO_LEGACY  1
#define ARL_PROMO_ASSOC   2

int         arl_abi_version(void);
const char* arl_last_error(void);

/* ------------------------------------------------------------------------- *
 * Return / advantage scans
 * ------------------------------------------------------------------------- */

/* GAE(lambda).  Replaces gen_adv_est, accel_rl/algos/pg/util.py:6-23, and the
 * per-env Python loop around it, accel_rl/algos/pg/aac_base.py:122-127.
 *   rewards, values  f32[n_env*horizon]   dones u8[n_env*horizon] (0/1)
 *   last_values      f32[n_env]           (bootstrap V, aac_base.py:112)
 *   advantages, returns  f32[n_env*horizon] (out; may not alias the inputs)   */
int arl_gae_scan(const float* rewards, const float* values, const uint8_t* dones,
                 const float* last_values, double discount, double gae_lambda,
                 int64_t n_env, int32_t horizon, int32_t promo,
                 float* advantages, float* returns, void* stream);

/* n-step discounted return + advantage.  Replaces discount_returns,
 * accel_rl/algos/pg/util.py:26-37, and `adv[:] = ret - v`, aac_base.py:115-121. */
int arl_nstep_return(const float* rewards, const uint8_t* dones, const float* values,
                     const float* last_values, double discount,
                     int64_t n_env, int32_t horizon, int32_t promo,
                     float* returns, float* advantages, void* stream);

/* valids mask + zeroing.  Replaces update_valids / zero_after_reset,
 * accel_rl/algos/pg/util.py:40-63 (loop at aac_base.py:129-134).
 *   reset_flags u8[n_env*horizon] = env_infos.need_reset if present else dones
 *   valids i8[n_env*horizon] (out); advantages/returns/values are zeroed IN
 *   PLACE after the first set flag (any of the three may be NULL).            */
int arl_valids_mask(const uint8_t* reset_flags, int64_t n_env, int32_t horizon,
                    int8_t* valids, float* advantages, float* returns, float* values,
                    void* stream);

/* (adv - mean) / (std + eps), population std, over all n samples or over the
 * valid ones.  Replaces accel_rl/algos/pg/aac_base.py:136-143.
 *   workspace: >= arl_standardize_workspace_bytes() bytes of device scratch.  */
int64_t arl_standardize_workspace_bytes(void);
int arl_standardize(float* advantages, const int8_t* valids_or_null, int64_t n,
                    double eps, void* workspace, void* stream);

/* ------------------------------------------------------------------------- *
 * Categorical action sampling
 * ------------------------------------------------------------------------- */

/* k = #{j : cumsum_j(prob[b,:]) < u[b]} clamped to n_actions-1; fp32 sequential
 * cumsum, fp64 compare.  Replaces weighted_sample_n, rllab/misc/special.py:22-27
 * (called from accel_rl/spaces/discrete.py:67-68 and
 * accel_rl/policies/pg/atari_cnn_policy.py:110).  The uniform variates the
 * reference draws with np.random.rand(B) are an explicit input.
 *   prob f32[batch*n_actions]  uniforms f64[batch]  actions u8[batch] (out)    */
int arl_sample_categorical(const float* prob, const double* uniforms,
                           int64_t batch, int32_t n_actions, uint8_t* actions,
                           void* stream);

/* ------------------------------------------------------------------------- *
 * Vectorised environment step (synthetic fixed-frame emulator + AtariEnv
 * wrapper + collector bookkeeping)
 * ------------------------------------------------------------------------- */

#define ARL_MAX_ACTIONS 18
#define ARL_TICKET_SHARDS 16
#define ARL_EPOCH_WORDS (32 * (ARL_TICKET_SHARDS + 1))
#define ARL_RAW_H 210
#define ARL_RAW_W 160
#define ARL_OBS_H 104     /* accel_rl/envs/atari_env.py:13 */
#define ARL_OBS_W 80

/* How the cropped 208 x 160 maximum of two raw frames becomes the 104 x 80 observation plane
 * (accel_rl/envs/atari_env.py:155: `cv2.resize(self._max_frame[:-2], (W, H), cv2.INTER_NEAREST)`).  In cv2's Python
 * signature resize(src, dsize[, dst[, fx[, fy[, interpolation]]]]) the constant lands in the `dst` slot, so what
 * RUNS is the default INTER_LINEAR, which for an exact 2x decimation is the rounded 2x2 box (a+b+c+d+2)>>2:
 * ARL_RESAMPLE_BOX2X, the default and the parity mode.  ARL_RESAMPLE_NEAREST is what the call NAMES (and what a
 * reference with the argument fixed would compute): dst(y, x) = src(2y, 2x), OpenCV's floor(dst * scale) rule.
 * OpenCV (opencv3=3.1.0, environment.yml:24) is not under /root/reference: both restated, parity unpinned.        */
#define ARL_RESAMPLE_BOX2X    0
#define ARL_RESAMPLE_NEAREST  1

/* Static description of one game + AtariEnv constructor arguments
 * (accel_rl/envs/atari_env.py:18-26).  Plain data, passed by pointer (host). */
typedef struct arl_game {
    const uint8_t* bank;        /* device u8[n_frames][210][160] frame bank      */
    int32_t n_frames;
    int32_t n_actions;
    int32_t action_set[ARL_MAX_ACTIONS]; /* ALE action codes (getMinimalActionSet)*/
    int32_t start_lives;
    int32_t life_period;
    int32_t frame_skip;         /* atari_env.py:20 */
    int32_t n_stack;            /* num_img_obs, atari_env.py:21 */
    int32_t clip_reward;        /* atari_env.py:22 */
    int32_t episodic_lives;     /* atari_env.py:23 */
    int32_t resample_mode;      /* ARL_RESAMPLE_*; atari_env.py:155 (0 = what the reference computes) */
} arl_game;

/* Per-env mutable state, struct-of-arrays; every pointer is device memory of
 * n_env elements unless noted.  Allocated and owned by the caller. */
typedef struct arl_env_state {
    int64_t  n_env;
    int32_t* tick;          /* emulator frames since reset_game                  */
    int32_t* emu_lives;     /* ale.lives()                                       */
    int32_t* env_lives;     /* AtariEnv._lives (atari_env.py:179)                */
    int32_t* phase;         /* per-emulator frame-bank phase                     */
    uint8_t* over;          /* ale.game_over()                                   */
    uint8_t* frozen;        /* NonResetCollector need_reset[i] (worker.py:75-95) */
    /* TrajInfo accumulators (accel_rl/sampler/util.py:75-101) */
    int32_t* traj_len;
    int32_t* traj_nonzero;
    float*   traj_ret;
    float*   traj_raw;
    float*   traj_disc;
    double*  traj_curdisc;
    /* per-step hand-off from arl_env_act_step to arl_env_frame_step */
    int32_t* frame_a;       /* bank index of raw_frame_1, -1 = all-zero frame    */
    int32_t* frame_b;       /* bank index of raw_frame_2                          */
    uint8_t* frame_mode;    /* 0 skip, 1 shift+push, 2 blank+push                 */
    uint8_t* reset_flag;    /* env must be reset by arl_env_frame_step            */
    /* start-noop streams: one per simulated worker process
     * (accel_rl/envs/atari_env.py:97 draws from the worker's numpy RNG)       */
    const uint8_t* noop_ring;   /* u8[n_streams][noop_ring_len] pre-drawn counts */
    int64_t* noop_cursor;       /* i64[2][n_streams], ping-pong by epoch parity  */
    int32_t* epoch;             /* i32[ARL_EPOCH_WORDS], zero-initialised: [0] number of env launches so far;
                                 * [2] arl_env_step's count of resets its one-launch-ahead forecast (next_reset) did
                                 * not announce -- must stay 0, a caller should check it once per batch;
                                 * the rest are arl_env_step's arrival tickets ([1] top, [32 (s + 1)] shard s: one
                                 * 128-byte line each when the array is 128-byte aligned) */
    int32_t  noop_ring_len;
    int32_t  envs_per_stream;
    /* completed-trajectory records (the reference's traj_infos_queue,
     * overlap/worker.py:147-148): appended with an atomic counter            */
    int32_t* done_count;        /* i32[1]                                        */
    int32_t* done_int;          /* i32[done_capacity][3] = env, Length, NonzeroRewards */
    float*   done_flt;          /* f32[done_capacity][3] = Return, RawReturn, DiscountedReturn */
    int32_t  done_capacity;
    /* arl_env_step only (both may be NULL for the two-launch path).
     * next_reset: u8[2][n_env], ping-pong by the parity of launch_count; [p][e] != 0 <=> env e will be flagged
     * for a mid-batch reset by its next step.  Written by arl_env_step and arl_env_reset for the launch that
     * follows; arl_env_act_step / arl_env_frame_step do not maintain it (after using them, reset every env with
     * arl_env_reset before the next arl_env_step).
     * launch_count: i32[1], arl_env_step / arl_env_reset launches on THIS state (the epoch above may be shared
     * with another state that draws from the same no-op streams, e.g. a worker's evaluation envs).          */
    uint8_t* next_reset;
    int32_t* launch_count;
} arl_env_state;

/* Rollout batch buffer, env-major (accel_rl/sampler/act_server/buffers.py:7-38).
 * Optional arrays may be NULL (raw_reward when !clip_reward, need_reset when
 * !episodic_lives: the reference's env_infos then lack the key).               */
typedef struct arl_rollout {
    int32_t  horizon;
    uint8_t* observations;  /* u8[n_env*horizon][n_stack][104][80]               */
    float*   rewards;       /* f32[n_env*horizon]                                */
    uint8_t* dones;         /* u8 (bool)                                         */
    float*   raw_reward;    /* env_infos.raw_reward                              */
    uint8_t* need_reset;    /* env_infos.need_reset                              */
    uint8_t* actions;       /* u8                                                */
    float*   prob;          /* agent_infos.prob f32[n_env*horizon][n_actions]    */
    float*   value;         /* agent_infos.value                                 */
    uint8_t* step_obs;      /* u8[n_env][n_stack][104][80] current observation   */
} arl_rollout;

/* One agent step for every env, scalar part (one lane per env): sample the
 * action, write actions/prob/value at index env*horizon+step, advance the
 * emulator frame_skip times, apply reward clipping / episodic-life / over-length
 * / reset rules, accumulate TrajInfo, write rewards/dones/env_infos.
 * Replaces: serve_actions' sample+scatter, overlap/sampler.py:139-145;
 * AtariEnv.step minus pixels, envs/atari_env.py:65-78,165-191;
 * ResetCollector / NonResetCollector.collect bookkeeping, overlap/worker.py:37-59,
 * 75-106; TrajInfo.step, sampler/util.py:92-101.
 *   prob f32[n_env][n_actions], value f32[n_env], uniforms f64[n_env] for THIS step
 *   active_or_null u8[n_env]: envs with 0 sit this step out untouched (used for the
 *   start-up decorrelation of sampler/util.py:34-57); NULL = every env steps    */
int arl_env_act_step(const arl_game* game, const arl_env_state* st, const arl_rollout* ro,
                     const float* prob, const float* value, const double* uniforms,
                     const uint8_t* active_or_null,
                     int32_t step, int32_t mid_batch_reset, double max_path_length,
                     double discount, void* stream);

/* Pixel part (one workgroup per env): resolve pending resets (start no-ops from
 * the env's stream, in env order within the stream), max of the two raw frames,
 * crop 2 rows, rounded 2x2 box to 104x80, shift/blank the frame stack, write
 * step_obs and observations[env*horizon + step + 1] (if step+1 < horizon).
 * Replaces AtariEnv._update_obs/_reset_obs/reset, envs/atari_env.py:93-100,
 * 151-163, and the observation writes of overlap/worker.py:51-53.
 *   max_start_noops: atari_env.py:24                                          */
int arl_env_frame_step(const arl_game* game, const arl_env_state* st, const arl_rollout* ro,
                       int32_t step, int32_t max_start_noops, void* stream);

/* arl_env_act_step + arl_env_frame_step as ONE launch (one workgroup per env; lane 0 does the scalar rules,
 * the workgroup the pixels, the last workgroup to finish advances the launch epoch): the env side of one agent
 * step of serve_actions / ResetCollector.collect, overlap/sampler.py:129-145, overlap/worker.py:37-59,75-106,
 * envs/atari_env.py:65-78,93-100,151-191.  Same arguments and results as the two calls in sequence.
 * Needs st->next_reset and max_path_length >= 1.
 *   single_write != 0 (needs mid_batch_reset != 0 and active_or_null == NULL): the new stacked observation is
 *   written once -- to observations[env*horizon + step + 1], or to step_obs after the last step of the batch --
 *   and the previous stack is read from observations[env*horizon + step] (which the caller has filled for
 *   step 0, overlap/worker.py:30-32); step_obs is then only current after the last step.  In this mode
 *   observations and step_obs must be 16-byte aligned (ARL_E_ALIGN otherwise; 8-byte alignment suffices without). */
int arl_env_step(const arl_game* game, const arl_env_state* st, const arl_rollout* ro,
                 const float* prob, const float* value, const double* uniforms,
                 const uint8_t* active_or_null, int32_t step, int32_t mid_batch_reset,
                 double max_path_length, double discount, int32_t max_start_noops,
                 int32_t single_write, void* stream);

/* Start of a batch: observations[env * horizon + 0] = step_obs[env] for every env (the collectors' first row,
 * overlap/worker.py:30-32) and st->done_count[0] = 0 (a fresh traj_infos queue), in one launch.                  */
int arl_rollout_begin(const arl_game* game, const arl_env_state* st, const arl_rollout* ro, void* stream);

/* Reset every env whose flag is set (u8[n_env]; NULL = all): start_envs with
 * max_decorrelation_steps == 0 (sampler/util.py:26-33) and
 * NonResetCollector.reset_needed_envs (overlap/worker.py:108-113, flags =
 * st->frozen, cleared afterwards).  Writes step_obs only.                      */
int arl_env_reset(const arl_game* game, const arl_env_state* st, const arl_rollout* ro,
                  const uint8_t* flags_or_null, int32_t max_start_noops, void* stream);

/* Stand-alone preprocess of explicit raw frame pairs (testing / other
 * emulators): out[i] = resample(crop(max(a[i], b[i]))), resample_mode = ARL_RESAMPLE_BOX2X (the reference's
 * arithmetic) or ARL_RESAMPLE_NEAREST; a may be NULL (zeros).
 * Replaces envs/atari_env.py:151-155.  a,b u8[n][210][160], out u8[n][104][80] */
int arl_preprocess_frames(const uint8_t* raw_a_or_null, const uint8_t* raw_b,
                          int64_t n, int32_t resample_mode, uint8_t* out, void* stream);

/* ------------------------------------------------------------------------- *
 * Learner side: minibatch gather and the flat-bucket optimiser step
 * ------------------------------------------------------------------------- */

/* dst[0 .. nbytes) = src[0 .. nbytes) as a KERNEL (capturable as a kernel node).  The one exception to "device
 * pointers only": either side may be PINNED host memory (hipHostMalloc / a pinned torch tensor), which the device
 * addresses directly -- this is how the per-batch host hand-offs (action uniforms, minibatch permutations, lr
 * multiplier in; episode records, gradient norms out: the reference's shared-memory step buffers and queues,
 * sampler/act_server/buffers.py:24-30, optimizers/util.py:8-18) enter and leave the hipGraphs without memcpy nodes.
 * A write to host memory is visible to the host once the stream has passed an event / synchronisation after it.   */
int arl_copy_bytes(void* dst, const void* src, int64_t nbytes, void* stream);
/* ring[(counter[0] % n_slots)][0 .. n) = src[0 .. n); counter[0] = (counter[0] + 1) % n_slots (kept reduced) -- the per-iteration diagnostics of an update
 * (the opt_infos of accel_rl/algos/pg/aac_base.py:104-106, e.g. GradNorm) leave a captured hipGraph into a slot the host
 * can name without a launch of its own (it counts the replays).  ring f32[n_slots][n], counter i32[1] on the device. */
int arl_ring_append(const float* src, int32_t n, float* ring, int32_t n_slots, int32_t* counter, void* stream);

/* out[b] = float(obs[idx[b]]) * scale  (u8 -> f32 gather; the reference gathers
 * on device with `s[idxs]`, accel_rl/optimizers/util.py:86-89, and scales by
 * 1/255 in ScalarFixedScaleLayer, accel_rl/policies/layers.py:22-41).
 *   obs u8[n_rows][row_bytes], idx i32[batch] (NULL = identity), out f32[batch][row_bytes] */
int arl_gather_scale_obs(const uint8_t* obs, const int32_t* idx_or_null, int64_t batch,
                         int64_t row_bytes, float scale, float* out, void* stream);

/* Channels-last variant of arl_gather_scale_obs for the conv stack:
 * obs u8[n_rows][channels][plane_bytes] -> out f32[batch][plane_bytes][channels].
 * channels must be 4 (the 4-frame stack, atari_env.py:21), plane_bytes % 16 == 0. */
int arl_gather_scale_obs_nhwc(const uint8_t* obs, const int32_t* idx_or_null, int64_t batch,
                              int32_t channels, int32_t plane_bytes, float scale, float* out,
                              void* stream);

/* x[rows][channels] = relu(x + bias[c]) in place: the bias + rectify of Lasagne's
 * Conv2DLayer / DenseLayer (accel_rl/policies/pg/networks/pg_cnn.py:47-68) on a
 * channels-last activation.  channels % 4 == 0.                                 */
int arl_bias_relu(float* x, const float* bias, int64_t rows, int32_t channels, void* stream);

/* Backward of the above: dy *= (y > 0) in place, dbias[c] = sum_rows dy (fixed
 * summation order).  workspace >= arl_relu_bwd_workspace_bytes().               */
int64_t arl_relu_bwd_workspace_bytes(void);
int arl_relu_bwd_bias_grad(float* dy, const float* y, int64_t rows, int32_t channels,
                           float* dbias, void* workspace, void* stream);
/* Deferred folds.  The weight-gradient and bias-gradient kernels write per-split partial sums; the
 * *_parts entry points stop there and describe the pending fold in *item (splits == 0: `out` is already
 * final), so that a whole backward pass ends in ONE arl_fold_many launch instead of one small fold
 * kernel per tensor.  Each item needs its own workspace region, live until arl_fold_many has run;
 * out[i] = sum_z part[z*total + i] in a fixed order (bit-reproducible). */
typedef struct arl_fold_item {
    const float* part;      /* f32[splits][total] */
    float*       out;       /* f32[total], 16-byte aligned */
    int64_t      total;     /* multiple of 4 */
    int32_t      splits;
    int32_t      valid;     /* > 0: `out` holds only this many floats (the partials are padded to `total`); 0 = total */
} arl_fold_item;
#define ARL_FOLD_MAX_ITEMS 24

/* Same, leaving the column-sum fold to arl_fold_many (see arl_fold_item). */
int arl_relu_bwd_bias_parts(float* dy, const float* y, int64_t rows, int32_t channels, float* dbias,
                            void* workspace, arl_fold_item* item, void* stream);

/* Policy / value heads + softmax for action serving: prob = softmax(h W_pi^T + b),
 * value = h w_v + b_v.  Replaces the output layers of _f_prob_value,
 * accel_rl/policies/pg/atari_cnn_policy.py:63-67 (pg_cnn.py:70-86).
 *   h f32[batch][hid]; w_head f32[n_actions+1][hid] (rows 0..A-1 pi, row A value);
 *   b_head f32[n_actions+1]; prob f32[batch][A]; value f32[batch]               */
int arl_pg_head_infer(const float* h, const float* w_head, const float* b_head, int64_t batch,
                      int32_t hid, int32_t n_actions, float* prob, float* value, void* stream);

/* Training-time heads: forward, the three losses and every gradient up to dh in
 * one pass.  kind 0 = A2C  pi_loss = -mean(log(pi[a]+1e-8) adv)    (a2c.py:43-46)
 *            kind 1 = PPO  pi_loss = -mean(min(r adv, clip(r, 1 -+ clip_param*lr_mult) adv)),
 *                          r = (pi[a]+1e-8)/(old[a]+1e-8)          (ppo.py:42-51)
 * v_loss = c_v mean((V-R)^2); ent_loss = -c_e mean(-sum pi log(pi+1e-8))
 * (aac_base.py:60-66, categorical.py:66-78); means are valids_mean when valids
 * is given (algos/pg/util.py:49-53; inv_count = 1/sum(valids) over the minibatch).
 * Rows of the batch arrays are selected by idx (NULL = identity).
 * tie_rule (PPO only) = how min() and clip() hand their gradient on (s1 = r adv, s2 = clip(r) adv, surr = min(s1, s2)):
 *   ARL_PPO_TIE_THEANO  the reference learner's graph as its Theano differentiates it.  accel_rl runs on
 *                       theano.gpuarray (runners/accel_rl_base.py:62-64; algos/dqn/cat_dqn.py:85-86 names
 *                       "Theano 0.9" / "1.0"), i.e. Theano >= 0.9; since 0.8 theano/scalar/basic.py has
 *                           Minimum.L_op:  e = eq(min, x);  gx = e gz;  gy = (1 - e) gz
 *                           ("This form handle the case when both value are the same. In that case, gx will be
 *                            gz, gy will be 0."; theano/tensor/tests/test_basic.py::test_maximum_minimum_grad:
 *                            "we only pass the gradient to the first input in that case")
 *                           Clip.L_op:     gx = ((x >= min) & (x <= max)) gz
 *                       and ppo.py:49 is T.minimum(surr_1, surr_2), so
 *                           d surr / d r = adv [surr == s1] + adv [surr != s1] [lo <= r <= hi]:
 *                       adv inside the clip range (the tie goes to the unclipped branch alone) and where s1 < s2
 *                       outside it, else 0.  Theano is a third-party dependency absent from /root/reference:
 *                       restated from its published source, parity unpinned.
 *   ARL_PPO_TIE_MATH    the mathematical derivative: adv inside the range, adv where s1 < s2 outside, else 0 (differs
 *                       from the above only where s1 == s2 by rounding OUTSIDE the range).
 *   ARL_PPO_TIE_BOTH    Theano <= 0.7 (gx = eq(min, x) gz, gy = eq(min, y) gz: a tie feeds BOTH arguments):
 *                       2 adv inside the clip range, bounds included; for comparing against runs of that vintage.
 *   out: dout f32[batch][A+1], dh f32[batch][hid] (before the hidden relu mask),
 *        dw_head f32[A+1][hid], db_head f32[A+1], loss4 f32[4] = pi, v, ent, pi+v+ent
 *   workspace >= arl_pg_head_workspace_bytes()                                  */
#define ARL_PPO_TIE_THEANO 0
#define ARL_PPO_TIE_MATH   1
#define ARL_PPO_TIE_BOTH   2
int64_t arl_pg_head_workspace_bytes(void);
int arl_pg_head_loss(const float* h, const float* w_head, const float* b_head,
                     const uint8_t* actions, const float* advantages, const float* returns,
                     const float* old_prob, const int8_t* valids_or_null,
                     const int32_t* idx_or_null, const float* lr_mult,
                     const float* inv_count_or_null, int64_t batch, int32_t hid, int32_t n_actions,
                     int32_t kind, int32_t tie_rule, float clip_param, float v_loss_coeff, float ent_loss_coeff,
                     int32_t relu_mask_dh, float* dout, float* dh, float* dw_head, float* db_head,
                     float* loss4, void* workspace, void* stream);
/* Same, stopping before the three small folds (dw_head, db_head, loss4): they are described in items3[0..2] for
 * arl_fold_many, so that a backward pass ends in ONE fold launch.  The bias partials are n_actions + 1 rounded
 * up to a multiple of 4 floats in the partials only (items3[1].valid = n_actions + 1); workspace stays live until the
 * fold has run.
 * wt_items_or_null / n_wt (ABI 4): the same launch also writes these layers' k-contiguous weight copies
 * (arl_conv2d_dgrad_weights below) in extra workgroups -- the backward pass that follows reads them, and this launch is
 * where a minibatch's parameters are final and the CUs are idle: one launch less per minibatch.                     */
int arl_pg_head_loss_parts(const float* h, const float* w_head, const float* b_head,
                           const uint8_t* actions, const float* advantages, const float* returns,
                           const float* old_prob, const int8_t* valids_or_null,
                           const int32_t* idx_or_null, const float* lr_mult,
                           const float* inv_count_or_null, int64_t batch, int32_t hid,
                           int32_t n_actions, int32_t kind, int32_t tie_rule, float clip_param,
                           float v_loss_coeff, float ent_loss_coeff, int32_t relu_mask_dh, float* dout, float* dh,
                           float* dw_head, float* db_head, float* loss4, void* workspace,
                           struct arl_fold_item* items3, const struct arl_dgrad_wt* wt_items_or_null, int32_t n_wt,
                           void* stream);

/* ------------------------------------------------------------------------- *
 * The policy network's dense contractions on the matrix cores (fp32 MFMA)
 * ------------------------------------------------------------------------- */

/* Geometry of one convolution layer; a dense layer is in_h = in_w = kh = kw = 1,
 * in_c = fan_in, out_c = units.  in_c and out_c must be multiples of 4.        */
typedef struct arl_conv_geom {
    int64_t batch;
    int32_t in_h, in_w, in_c;     /* input  x  f32[batch][in_h][in_w][in_c]  (NHWC)     */
    int32_t out_c, kh, kw;        /* weight w  f32[out_c][kh][kw][in_c] (correlation)   */
    int32_t stride, pad_h, pad_w; /* output y  f32[batch][out_h][out_w][out_c],
                                     out_h = (in_h + 2 pad_h - kh) / stride + 1          */
    int32_t route;                /* how the fp32 contractions of this call are computed: ARL_CONV_ROUTE_*  */
} arl_conv_geom;

/* arl_conv_geom::route (the reference's floatX is float32: accel_rl/policies/pg/networks/pg_cnn.py:45-86 through
 * Theano).  Operands and results are fp32 on every route; only the way through the matrix cores differs:
 *   ARL_CONV_ROUTE_SPLIT9 (0, the default of a zero-initialised struct): each fp32 operand is split EXACTLY into three
 *      bf16 pieces (24 significand bits = 3 x 8) and all nine piece products -- each exact in fp32 -- are accumulated
 *      in fp32 by v_mfma_f32_32x32x16_bf16: every product term of the fp32 contraction enters the sum exactly, only
 *      the accumulation rounds;
 *   ARL_CONV_ROUTE_FP32   v_mfma_f32_32x32x2_f32: bit for bit a k-ordered fmaf chain (157 TF/s peak on gfx950);
 *   ARL_CONV_ROUTE_SPLIT6 as SPLIT9 without the three smallest piece products (each below 2^-24 of |x y|);
 *   ARL_CONV_ROUTE_BF16   NOT an fp32 contraction -- the labelled reduced-precision option: each fp32 operand is ROUNDED
 *      (to nearest even) to one bf16 value on its way into the matrix cores, one product per multiply, fp32
 *      accumulation; tensors in memory stay fp32.  8 significand bits per operand: results differ from the other
 *      routes by ~2^-9 relative per product; never a default, never selected by the library.
 * u8 observations are exact in one bf16 piece (three products on both split routes, one on BF16).  Layers with <= 16 output
 * columns and the generic (any channel count) kernels always take the fp32 chain -- except the first convolution from u8
 * rows with 16 filters of 8 x 8 (spec 0), which runs on the image-stationary bf16-split kernel with half its tile idle.  Deterministic on every route;
 * any other value: ARL_E_ARG.  The route is an argument of the call: the library keeps no mode.                     */
#define ARL_CONV_ROUTE_SPLIT9 0
#define ARL_CONV_ROUTE_FP32   1
#define ARL_CONV_ROUTE_SPLIT6 6
#define ARL_CONV_ROUTE_BF16   2

/* Scratch for the split reductions below (fixed; the caller allocates once). */
int64_t arl_conv_workspace_bytes(void);

/* y = conv(x, w) + bias, then max(., 0) if relu.  Replaces the forward of Lasagne's
 * Conv2DLayer / DenseLayer as used by PgCnn (accel_rl/policies/pg/networks/pg_cnn.py:47-68,
 * policies/layers.py:22-41; the reference's flipped filters are stored pre-flipped).
 * Deterministic: fp32 MFMA accumulation in k order, split-K folded in a fixed order. */
int arl_conv2d_fwd(const float* x, const float* w, const float* bias_or_null, float* y,
                   const arl_conv_geom* geom, int32_t relu, void* workspace, void* stream);

/* dx = gradient of the layer input given dy (every element of dx is written).
 * If mask is given (same shape as dx): dx = 0 where mask <= 0 -- the rectifier
 * backward of the previous layer.  Requires kh % stride == 0 and kw % stride == 0.
 * Replaces the T.grad of the same layers (optimizers/single/ppo_optimizer.py:38-40). */
/* job (optional): an optimiser job (arl_corun_job, below) that this call's launch may carry in extra workgroups;
 * *job_taken = 1 if it did (only the scalar-addressed data-gradient launches of layers with > 16 input channels
 * can), else 0 and the caller runs the job itself (arl_corun_job_run). */
struct arl_corun_job;
int arl_conv2d_bwd_data(const float* dy, const float* w, const float* wt_or_null, const float* mask_or_null, float* dx,
                        const arl_conv_geom* geom, const struct arl_corun_job* job_or_null,
                        int32_t* job_taken_or_null, void* stream);

/* The data gradient's own copy of a layer's weights (ABI 4): per input-pixel parity class (ph, pw) of the stride a matrix
 * wt[ph * stride + pw][in_c][(ty * kw / stride + tx) * out_c + k] = w[k][i0 + stride ty][j0 + stride tx][c], (i0, j0) =
 * ((ph + pad_h) % stride, (pw + pad_w) % stride) -- the reduction index of dx = conv^T(dy, w) contiguous, as a forward pass
 * finds its weights.  Same size as w; one launch converts up to ARL_DGRAD_WT_MAX layers (after every parameter update,
 * before the backward pass that reads them).  Given as wt_or_null to arl_conv2d_bwd_data / arl_conv2d_bwd_pair, the
 * bf16-split kernels of 17 .. 64 input channels read it instead of w: the same piece products in the same order
 * (bit-identical results), a third less LDS and loader work per k-tile.  The reference has no counterpart: Theano's
 * conv gradient picks its own layout inside cuDNN (T.grad of pg_cnn.py:47-68, optimizers/single/ppo_optimizer.py:38-40). */
typedef struct arl_dgrad_wt {
    const float* w;             /* f32[out_c][kh][kw][in_c]                                                */
    float* wt;                  /* f32[out_c * kh * kw * in_c], 16-byte aligned, not aliasing w            */
    const arl_conv_geom* geom;  /* kh, kw divisible by stride, stride <= 2                                 */
} arl_dgrad_wt;
#define ARL_DGRAD_WT_MAX 4
int arl_conv2d_dgrad_weights(const arl_dgrad_wt* items, int32_t n, void* stream);

/* dw f32[out_c][kh][kw][in_c] = gradient of the layer weights given dy and the layer
 * input x; the reduction over batch x out_h x out_w is split across workgroups and
 * folded in a fixed order (no atomics).                                           */
int arl_conv2d_bwd_weight(const float* dy, const float* x, float* dw, const arl_conv_geom* geom,
                          void* workspace, void* stream);

/* Deferred-fold variant (arl_fold_item above). */
/* dbias (optional): the kernel also leaves per-split column sums of dy (the bias gradient of a layer
 * whose dy is already masked by its rectifier) behind the weight partials and describes their fold in
 * *bias_item; bias_item->splits == -1 means "not produced" (the generic kernels ran): use
 * arl_relu_bwd_bias_grad / _parts instead. */
int arl_conv2d_bwd_weight_parts(const float* dy, const float* x, float* dw, const arl_conv_geom* geom,
                                void* workspace, int64_t workspace_bytes, arl_fold_item* item,
                                float* dbias_or_null, arl_fold_item* bias_item_or_null, void* stream);
int arl_fold_many(const arl_fold_item* items, int32_t n, void* stream);

/* Forward with the split reduction left unfolded: as arl_conv2d_fwd, but when the launch split its reduction the partial
 * sums stay in `workspace` and *item describes them (part f32[splits][rows * out_c], total = rows * out_c; bias and
 * rectifier NOT applied: they belong to whoever folds -- arl_env_step_served below does, per env, inside its launch);
 * item->splits == 0: the launch did not split, y is final (bias and rectifier applied).  The workspace stays live until
 * the consumer has run.  Replaces the same Lasagne DenseLayer forward as arl_conv2d_fwd (pg_cnn.py:57-68).          */
int arl_conv2d_fwd_parts(const float* x, const float* w, const float* bias_or_null, float* y,
                         const arl_conv_geom* geom, int32_t relu, void* workspace, arl_fold_item* item, void* stream);

/* ------------------------------------------------------------------------- *
 * One agent step of action serving in ONE launch
 * ------------------------------------------------------------------------- */

/* The policy's output layers as arl_env_step_served evaluates them for every env (row e of each array = env e). */
typedef struct arl_serve_head {
    arl_fold_item hidden;       /* the last hidden layer as arl_conv2d_fwd_parts left it: total = n_env * hid;
                                 * splits > 0: part f32[splits][n_env][hid] partial sums (folded in arl_fold_many's order);
                                 * splits == 0: part f32[n_env][hid] finished activations.  `out` is not used.          */
    const float* hidden_bias;   /* f32[hid] or NULL: added after the fold (splits > 0 only)                             */
    int32_t hidden_relu;        /* != 0: max(., 0) after the bias (splits > 0 only)                                     */
    int32_t hid;                /* multiple of 4, <= 1024                                                               */
    const float* w_head;        /* f32[n_actions + 1][hid], rows 0..A-1 pi, row A value (arl_pg_head_infer's layout)    */
    const float* b_head;        /* f32[n_actions + 1]                                                                   */
} arl_serve_head;

/* The first convolution of the NEXT observation, evaluated from LDS right after the env step has built it
 * (arl_conv2d_u8_fwd's arithmetic; geometries: arl_serve_conv1_supported).                                            */
typedef struct arl_serve_conv1 {
    const arl_conv_geom* geom;  /* batch = n_env, in_c = n_stack, 104 x 80 input, 32 or 16 filters of 8 x 8, no padding*/
    const float* w;             /* f32[out_c][n_stack][8][8]                                                           */
    const float* bias;          /* f32[out_c] or NULL                                                                  */
    float* y;                   /* f32[n_env][out_h][out_w][out_c]                                                     */
    float scale;                /* pixel scale (1 / 255), applied to the finished sums                                 */
    int32_t relu;
} arl_serve_conv1;

/* 1 if arl_env_step_served can take this first layer (else: conv1_or_null = NULL and arl_conv2d_u8_fwd afterwards). */
int arl_serve_conv1_supported(const arl_game* game, const arl_conv_geom* geom);

/* arl_rollout_begin and arl_conv2d_u8_fwd of the rows it copies as ONE launch: observations[env * horizon + 0] =
 * step_obs[env] (overlap/worker.py:30-32), st->done_count[0] = 0, and conv1->y = the first convolution of those rows
 * (pg_cnn.py:47-52) -- the image passes through the kernel's registers once.  Geometries: arl_serve_conv1_supported.  */
int arl_rollout_begin_conv1(const arl_game* game, const arl_env_state* st, const arl_rollout* ro,
                            const arl_serve_conv1* conv1, void* stream);

/* One (step, all envs) turn of serve_actions with everything per-env in one launch (one workgroup per env): fold, bias
 * and rectifier of the last hidden layer's split partials, the policy and value heads + softmax (arl_pg_head_infer),
 * weighted_sample_n, the env step (arl_env_step with mid_batch_reset != 0, single_write != 0, every env stepping) and,
 * with conv1, the first convolution of the observation the step has just produced (row e of conv1->y = env e; it
 * belongs to step + 1, or to the bootstrap observation after the batch's last step).  prob and value are written
 * straight to ro->prob / ro->value rows env * horizon + step.  Results are bit for bit those of the separate calls.
 * Replaces accel_rl/sampler/act_server/alternating/overlap/sampler.py:120-151 (serve_actions), the output layers of
 * _f_prob_value (policies/pg/atari_cnn_policy.py:63-67, pg/networks/pg_cnn.py:57-86), rllab/misc/special.py:22-27,
 * overlap/worker.py:37-59, envs/atari_env.py:65-78,93-100,151-191 and pg_cnn.py:47-52 for the next observation.
 * Needs st->next_reset, max_path_length >= 1, n_stack <= 4, 16-byte aligned observations / step_obs.               */
int arl_env_step_served(const arl_game* game, const arl_env_state* st, const arl_rollout* ro,
                        const arl_serve_head* head, const arl_serve_conv1* conv1_or_null,
                        const double* uniforms, int32_t step, double max_path_length, double discount,
                        int32_t max_start_noops, void* stream);

/* A layer's data gradient and weight gradient (deferred fold) as ONE launch: the two are independent
 * and both read dy, so their workgroups share a grid -- one ramp-up and one tail instead of two, and
 * the second problem's workgroups fill the CUs the first one's last wave leaves idle.  Same results as
 * arl_conv2d_bwd_data + arl_conv2d_bwd_weight_parts (which it falls back to when either side is not
 * on the scalar-addressed fast path).  wt_or_null: arl_conv2d_dgrad_weights' copy of w for the data gradient. */
int arl_conv2d_bwd_pair(const float* dy, const float* w, const float* wt_or_null, const float* mask_or_null, float* dx,
                        const float* x, float* dw, const arl_conv_geom* geom, void* workspace,
                        int64_t workspace_bytes, arl_fold_item* item, float* dbias_or_null,
                        arl_fold_item* bias_item_or_null, const struct arl_corun_job* job_or_null,
                        int32_t* job_taken_or_null, void* stream);

/* Convolution 1 read straight from the sampler's observations (no f32 copy of the input):
 * obs u8[obs_rows][in_c][in_h][in_w] (the layout of samples_buf.observations,
 * accel_rl/sampler/act_server/buffers.py:7-38), row b of the batch = obs[idx ? idx[b] : b]
 * (the minibatch indices of optimizers/util.py:8-18), x = float(byte) * scale
 * (atari_cnn_policy.py:88-91: the network input is obs * (1 / 255)).  Weights and their gradient are
 * f32[out_c][in_c][kh][kw] (correlation kernels).  geom->batch = rows of the batch; in_c any count >= 1.
 * Requires pad 0, stride % 4 == 0, in_w % 4 == 0, (in_h * in_w) % 4 == 0, kw in {4, 8, 16},
 * kh % (16 / kw) == 0, out_c % 4 == 0 and <= 32 (else ARL_E_RANGE: use arl_gather_scale_obs_nhwc +
 * arl_conv2d_fwd).  The sums run over the exact integer pixels, plane by plane, and `scale` multiplies the
 * finished sum once (a convolution is linear in its input: y = scale * conv(byte, w) + bias) -- equal to the
 * gather + scale route up to f32 round-off, not bit for bit.  */
int arl_conv2d_u8_fwd(const uint8_t* obs, int64_t obs_rows, const int32_t* idx_or_null, float scale,
                      const float* w, const float* bias_or_null, float* y, const arl_conv_geom* geom,
                      int32_t relu, void* stream);
/* Its weight gradient (deferred fold, optional bias-gradient partials: as arl_conv2d_bwd_weight_parts). */
int arl_conv2d_u8_bwd_weight_parts(const float* dy, const uint8_t* obs, int64_t obs_rows,
                                   const int32_t* idx_or_null, float scale, float* dw,
                                   const arl_conv_geom* geom, void* workspace, int64_t workspace_bytes,
                                   arl_fold_item* item, float* dbias_or_null,
                                   arl_fold_item* bias_item_or_null, void* stream);

/* ------------------------------------------------------------------------- *
 * Replay memory of the DQN family
 * ------------------------------------------------------------------------- */

#define ARL_REPLAY_MAX_HORIZON 16

/* Frame-dedup replay storage for all environments, struct-of-arrays in HBM.
 * Replaces FrameReplayBuffer + one EnvBuffer per environment,
 * accel_rl/algos/dqn/replay_buffers/frame.py:23-119.  Per environment `size`
 * states; every frame is stored once in a ring of size + n_stack - 1 slots. */
typedef struct arl_replay {
    int64_t  n_env;
    int32_t  size;            /* states per environment (env_replay_size, frame.py:44) */
    int32_t  n_stack;         /* frames per observation (num_img_obs), >= 2            */
    int32_t  frame_bytes;     /* bytes of one frame, multiple of 16 (104*80 = 8320)    */
    int32_t  reward_horizon;  /* n of the n-step return                                */
    uint8_t* frames;          /* u8[n_env][size + n_stack - 1][frame_bytes]            */
    uint8_t* n_blanks;        /* u8[n_env][size + n_stack - 1] blank frames after a reset */
    uint8_t* acts;            /* u8[n_env][size]                                       */
    uint8_t* terminals;       /* u8[n_env][size] (0/1)                                 */
    float*   rewards;         /* f32[n_env][size]                                      */
    float*   returns;         /* f32[n_env][size] n-step discounted return             */
} arl_replay;

/* One sampler batch into the ring at state index idx: newest frame of every step,
 * actions / rewards / dones, blank-history marks after terminals, and the n-step
 * returns of the `horizon` states that now have all their rewards, with a terminal
 * inside the window propagated back.  Replaces append_data / write_samples,
 * frame.py:57-60,121-166.  Sampler layout, env-major: observations
 * u8[n_env*horizon][n_stack][frame_bytes], actions u8, rewards f32, dones u8.
 * promo as for the scans (the reference accumulates python-float x float32). */
int arl_replay_append(const arl_replay* rb, const uint8_t* observations, const uint8_t* actions,
                      const float* rewards, const uint8_t* dones, int32_t horizon, int32_t idx,
                      double discount, int32_t promo, void* stream);

/* Batch extraction: obs / next_obs (reward_horizon states later) as stacked u8 frames with
 * the post-reset blank frames zeroed, plus actions, n-step returns, terminals.
 * Replaces extract_batch / extract_observations, frame.py:69-90. */
int arl_replay_extract(const arl_replay* rb, const int32_t* env_idxs, const int32_t* step_idxs,
                       int64_t batch, uint8_t* obs, uint8_t* next_obs, uint8_t* actions,
                       float* returns, uint8_t* terminals, void* stream);

/* Parted sum tree of prioritized replay, f64[2^levels - 1], root at 0, children 2i+1 / 2i+2
 * (accel_rl/algos/dqn/replay_buffers/sum_tree.py:12-98).
 * find:   descend by prefix mass, uniforms in [0,1] scaled by the root       (:88-98)
 * add:    np.add.at along every leaf-to-root path; several updates of one node are
 *         applied in INPUT order, so the f64 rounding equals the reference's (:54-57)
 * gather: out[i] = scale * tree[idxs[i]]                                      (:65,:83) */
int arl_sumtree_find(const double* tree, int32_t levels, const double* uniforms, int64_t n,
                     int32_t* tree_idxs, void* stream);
int arl_sumtree_add(double* tree, int32_t levels, const int32_t* tree_idxs, const double* diffs,
                    int64_t n, void* stream);
int arl_sumtree_gather(const double* tree, const int32_t* tree_idxs, int64_t n, double scale,
                       double* out, void* stream);

/* The device half of PartedSumTree.sample_n (sum_tree.py:77-86): find() for m uniforms, sorted distinct
 * leaves, the n smallest kept with their probabilities and (part, step) = divmod(leaf, part_size);
 * n_unique[0] = number of distinct leaves found (< n: the reference would draw more -- the caller tops
 * up through arl_sumtree_find as before; output slots past the distinct leaves repeat the first one, so
 * consumers already queued stay in bounds).  1 <= n <= m <= 4096.                                 */
int arl_sumtree_sample(const double* tree, int32_t levels, const double* uniforms, int32_t m, int32_t n,
                       int32_t part_size, int32_t* tree_idxs, int32_t* env_idxs, int32_t* step_idxs,
                       double* probs, int32_t* n_unique, void* stream);

/* The same launch with the batch's importance-sampling weights (arl_is_weights on the n probabilities: prioritized.py:33-35)
 * and a host hand-off without a copy: uniforms may live in page-locked host memory (read once each), and
 * notify_or_null (page-locked, 8-byte aligned) receives (ticket << 32) | n_unique when everything above is written --
 * the one integer sample_n's caller waits for (sum_tree.py:80: `while len(tree_idxs) < n`) arrives by a store the host
 * polls instead of a memcpy node and an event.  is_weights_or_null f32[n] (slots past the distinct leaves: 0).       */
int arl_sumtree_sample_batch(const double* tree, int32_t levels, const double* uniforms, int32_t m, int32_t n,
                             int32_t part_size, int32_t* tree_idxs, int32_t* env_idxs, int32_t* step_idxs,
                             double* probs, int32_t* n_unique, double beta, float* is_weights_or_null,
                             int64_t* notify_or_null, int32_t ticket, void* stream);

/* update_batch_priorities on the device in ONE launch: arl_priority_diffs (f32 priorities ** alpha - the probabilities
 * sampled before) feeding arl_sumtree_add (np.add.at per level in input order); prioritized.py:37-38,
 * sum_tree.py:50-57,74-75.                                                                                          */
int arl_sumtree_update_pow(double* tree, int32_t levels, const int32_t* tree_idxs, const float* priorities,
                           const double* last_probs, double alpha, int64_t n, void* stream);

/* Importance-sampling weights of PrioritizedReplayBuffer.sample_batch (prioritized.py:33-35):
 * out[i] = f32( (1 / probs[i]) ** beta / max_j (1 / probs[j]) ** beta ), arithmetic in f64.        */
int arl_is_weights(const double* probs, int64_t n, double beta, float* out, void* stream);

/* diffs[i] = f64( f32(priorities[i] ** alpha) ) - last_probs[i]: the argument update_last_samples hands to
 * reconstruct (prioritized.py:37-38, sum_tree.py:74-75); follow with arl_sumtree_add.                */
int arl_priority_diffs(const float* priorities, const double* last_probs, int64_t n, double alpha,
                       double* diffs, void* stream);

/* ------------------------------------------------------------------------- *
 * Categorical DQN output stage
 * ------------------------------------------------------------------------- */

/* Action serving: per-action softmax over atoms, Q_a = sum_i p_ai z_i, greedy action = first
 * maximum; where override[b] >= 0 that action is taken instead (the epsilon-greedy draw, made
 * on the host RNG as the reference does).  The chosen action is written as a one-hot row so the
 * sampler's categorical kernel selects exactly it.  Replaces AtariCatDqnPolicy.get_actions /
 * actions_sym, accel_rl/policies/dqn/atari_cat_dqn_policy.py:84-126 (+ catdqn_cnn.py:94-99).
 *   logits f32[batch][n_actions][atom_stride], atom_stride % 4 == 0 >= n_atoms <= 64;
 *   z f32[n_atoms]; onehot f32[batch][n_actions]; greedy u8[batch] or NULL
 * dueling != 0 (DuelingMergeLayer, policies/dqn/layers/dueling_merge_layer.py:32-35, catdqn_cnn.py:77-93):
 *   logits f32[batch][n_actions + 1][atom_stride], advantage rows then ONE value row;
 *   logit(a, i) = val_i + (adv_ai - mean_a adv_ai) before the softmax.                    */
int arl_catdqn_act(const float* logits, const float* z, const int32_t* override_or_null, int64_t batch,
                   int32_t n_actions, int32_t n_atoms, int32_t atom_stride, int32_t dueling, float* onehot,
                   uint8_t* greedy_or_null, void* stream);

/* Loss of CategoricalDQN.build_loss, accel_rl/algos/dqn/cat_dqn.py:40-109: next action greedy
 * under the target net (or the policy net: double DQN), its atom probabilities under the target
 * net projected from the support shifted by the n-step return (clipped to [v_min, v_max], zeroed
 * gamma^n z where terminal) onto the base support; cross-entropy against clip(pred, 1e-6, 1),
 * importance-weighted mean; priorities = clip(KL, 1e-6, 1e6).
 *   out: dlogits f32[batch][n_actions][atom_stride] (d mean-loss / d pred_logits),
 *        loss_rows f32[batch] (their sum is the loss), kl f32[batch]
 * dueling != 0: all three logit blocks and dlogits are [n_actions + 1][atom_stride] as above;
 *   dlogits is the gradient w.r.t. the advantage rows and the value row (through the merge). */
int arl_catdqn_loss(const float* pred_logits, const float* tgt_next_logits, const float* pol_next_logits_or_null,
                    const float* z, const uint8_t* actions, const float* returns, const uint8_t* terminals,
                    const float* is_weights_or_null, int64_t batch, int32_t n_actions, int32_t n_atoms,
                    int32_t atom_stride, int32_t dueling, float v_min, float v_max, float gamma_n,
                    float* dlogits, float* loss_rows, float* kl, void* stream);

/* The same loss reading its three logit blocks as the output layer's SPLIT PARTIAL SUMS (arl_conv2d_fwd_parts on the
 * "action_atoms" dense layer, catdqn_cnn.py:69-76): at the reference's minibatch of 32 (accel_rl/algos/dqn/dqn.py:18) an
 * update is a chain of launch latencies, and the two launches that only fold the output layers' partials are taken
 * over by the loss kernel -- logit = (the partials summed in arl_fold_many's order) + bias, operation for operation, so
 * the results are arl_catdqn_loss's on the folded logits bit for bit.
 *   part          f32: this block's row 0 inside split 0 (the online pass over [obs; next_obs] hands `pred` its first
 *                 batch rows and `pol_next` the rows from batch on, same split_stride)
 *   split_stride  floats between consecutive splits (arl_fold_item.total of the forward launch)
 *   splits        1 .. 127 (arl_fold_item.splits; a launch that did not split: 1, with part = its finished output and
 *                 bias_or_null = NULL)
 *   bias_or_null  f32[(n_actions (+ 1)) * atom_stride]: the output layer's bias, added after the sum                  */
typedef struct arl_logit_src {
    const float* part;
    const float* bias_or_null;
    int64_t split_stride;
    int32_t splits;
    int32_t reserved;
} arl_logit_src;
int arl_catdqn_loss_parts(const arl_logit_src* pred, const arl_logit_src* tgt_next, const arl_logit_src* pol_next_or_null,
                          const float* z, const uint8_t* actions, const float* returns, const uint8_t* terminals,
                          const float* is_weights_or_null, int64_t batch, int32_t n_actions, int32_t n_atoms,
                          int32_t atom_stride, int32_t dueling, float v_min, float v_max, float gamma_n,
                          float* dlogits, float* loss_rows, float* kl, const struct arl_dgrad_wt* wt_items_or_null,
                          int32_t n_wt, void* stream);
/* (wt_items_or_null / n_wt: as in arl_pg_head_loss_parts -- the launch also writes these layers' k-contiguous weight copies
 *  for the backward pass that follows, in extra workgroups.) */

/* Plain DQN action serving: greedy action = first maximum of the Q row (T.argmax), override as
 * above, one-hot row out.  Replaces AtariDqnPolicy.get_actions / actions_sym,
 * accel_rl/policies/dqn/atari_dqn_policy.py:61-63,76-79,118-130.
 *   q f32[batch][q_stride], q_stride % 4 == 0 >= n_actions (<= 255); onehot f32[batch][n_actions]
 * dueling != 0 (dqn_cnn.py:89-112): columns 0..n_actions-1 are advantages, column n_actions the
 *   value; q_a = val + (adv_a - mean adv).                                                  */
int arl_dqn_act(const float* q, const int32_t* override_or_null, int64_t batch, int32_t n_actions,
                int32_t q_stride, int32_t dueling, float* onehot, uint8_t* greedy_or_null, void* stream);

/* Loss of DQN.build_loss, accel_rl/algos/dqn/dqn.py:137-172: next_q = max_a target(next_obs) or
 * (double DQN) target(next_obs)[argmax_a policy(next_obs)]; y = return + (1 - terminal) gamma^n
 * next_q; d = y - q[action]; 0.5 d^2, or the Huber loss with threshold delta_clip (> 0);
 * importance-weighted mean; priorities = clip(|d|, 0, delta_clip) (|d| when delta_clip <= 0).
 *   out: dq f32[batch][q_stride] (d mean-loss / d q; zero outside the taken action),
 *        loss_rows f32[batch] (their sum is the loss), td_abs f32[batch]
 * dueling != 0: rows as in arl_dqn_act; dq is the gradient w.r.t. advantages and value.        */
int arl_dqn_loss(const float* q, const float* tgt_next_q, const float* pol_next_q_or_null,
                 const uint8_t* actions, const float* returns, const uint8_t* terminals,
                 const float* is_weights_or_null, int64_t batch, int32_t n_actions, int32_t q_stride,
                 int32_t dueling, float gamma_n, float delta_clip, float* dq, float* loss_rows, float* td_abs,
                 void* stream);

/* ------------------------------------------------------------------------- *
 * LSTM cell of the recurrent policies
 * ------------------------------------------------------------------------- */

/* Elementwise part of FastLstmLayer.step, accel_rl/policies/layers.py:331-346: gate order
 * f, i, c~, o; f, i, o = sigmoid, c~ = tanh; c = f c_prev + i c~; h = o tanh(c).  gx = x W_x + b and
 * gh = h_prev W_h are the callers' dense products (gh may be NULL = zero).  Every *_stride is the
 * element distance between consecutive rows, so a time slice of a [trajectory][time] batch can be
 * addressed in place.  gates (optional) receives the activated gates for the backward pass. */
int arl_lstm_cell_fwd(const float* gx, int64_t gx_stride, const float* gh_or_null, const float* c_prev,
                      int64_t cprev_stride, int64_t batch, int32_t hidden, float* h_out, int64_t h_stride,
                      float* c_out, int64_t c_stride, float* gates_or_null, int64_t gates_stride, void* stream);

/* Backward of the above for one time step: dh (from the layers above, strided) + dh_rec (from step
 * t+1, contiguous) and dc_next -> pre-activation gate gradients dgates[B][4H] and dc_prev. */
int arl_lstm_cell_bwd(const float* dh_or_null, int64_t dh_stride, const float* dh_rec_or_null,
                      const float* dc_next_or_null, const float* gates, int64_t gates_stride,
                      const float* c_prev, int64_t cprev_stride, const float* c_out, int64_t c_stride,
                      int64_t batch, int32_t hidden, float* dgates, int64_t dgates_stride, float* dc_prev,
                      void* stream);

/* GRU cell (GruLayer.step, accel_rl/policies/layers.py:163-168), gate order r, u, c in the 3H-wide
 * arrays: r = s(gx_r + gh_r); u = s(gx_u + gh_u); c = tanh(gx_c + r gh_c); h = (1 - u) h_prev + u c.
 * gx = x [W_xr W_xu W_xc] + b, gh = h_prev [W_hr W_hu W_hc] are the callers' dense products.
 * saved (optional, [B][4H]) receives r, u, c, gh_c for the backward pass. */
int arl_gru_cell_fwd(const float* gx, int64_t gx_stride, const float* gh, const float* h_prev,
                     int64_t hprev_stride, int64_t batch, int32_t hidden, float* h_out, int64_t h_stride,
                     float* saved_or_null, int64_t saved_stride, void* stream);

/* Backward of one GRU step: dh (layers above, strided) + dh_rec + dh_dir (both contiguous, from
 * step t+1) -> dgx[B][3H] (gradient wrt gx), dgh[B][3H] (wrt gh; its c block carries the factor r)
 * and dh_prev[B][H] = the direct part dh (1 - u); the caller adds dgh W_h^T. */
int arl_gru_cell_bwd(const float* dh_or_null, int64_t dh_stride, const float* dh_rec_or_null,
                     const float* dh_dir_or_null, const float* saved, int64_t saved_stride,
                     const float* h_prev, int64_t hprev_stride, int64_t batch, int32_t hidden,
                     float* dgx, int64_t dgx_stride, float* dgh, int64_t dgh_stride, float* dh_prev,
                     void* stream);

/* Plain recurrent cell (RecurrentLayer.step, layers.py:80-82): h = tanh(gx + gh), and its backward
 * dpre = (dh + dh_rec) (1 - h^2). */
int arl_rnn_cell_fwd(const float* gx, int64_t gx_stride, const float* gh, int64_t batch, int32_t hidden,
                     float* h_out, int64_t h_stride, void* stream);
int arl_rnn_cell_bwd(const float* dh_or_null, int64_t dh_stride, const float* dh_rec_or_null,
                     const float* h_out, int64_t h_stride, int64_t batch, int32_t hidden, float* dpre,
                     int64_t dpre_stride, void* stream);

/* Optimiser state for ONE flat fp32 parameter bucket (all trainable params in
 * get_params order, accel_rl/optimizers/util.py:35-39). */
typedef struct arl_opt_state {
    int64_t n_params;
    float*  params;         /* f32[P] flat parameter vector (updated in place)   */
    float*  grads;          /* f32[P] flat gradient (after all-reduce in sync mode) */
    float*  slot0;          /* adam m / rmsprop accu                             */
    float*  slot1;          /* adam v / unused                                   */
    float*  step_count;     /* f32[1] Lasagne's t (floatX)                       */
    float*  lr_mult;        /* f32[1] device scalar (linear schedule, aac_base.py:165-168) */
    double* partials;       /* f64[ARL_OPT_PARTIALS] scratch                     */
    float*  grad_norm_log;  /* f32[norm_log_len] ring: norm of update k at k % len */
    int32_t norm_log_len;
} arl_opt_state;

#define ARL_OPT_PARTIALS 1024
#define ARL_OPT_ADAM     0
#define ARL_OPT_RMSPROP  1

/* grads *= avg_factor; norm = ||grads||_2; if clip > 0: grads *= clip(norm,0,clip)/(1e-7+norm);
 * then adam / rmsprop.  Two launches (sum of squares, fused update).
 * Replaces avg_grads_from_flat + apply_grad_norm_clip + lasagne update,
 * accel_rl/optimizers/util.py:63-76, sync/sync_ppo_optimizer.py:27-34; update
 * arithmetic as in accel_rl/optimizers/update_methods_stats.py:11-33 (rmsprop)
 * and :55-87 (adam).  clip <= 0 means "no clip" (norm still logged).          */
int arl_opt_step(const arl_opt_state* opt, int32_t method, float learning_rate,
                 float avg_factor, float clip, float beta1_or_rho, float beta2,
                 float epsilon, void* stream);

/* The same update WITHOUT norm clipping (PPO's default, accel_rl/algos/pg/ppo.py:24: grad_norm_clip=None) as ONE
 * launch: the sum of squares for the logged norm is taken in the update's own pass over the gradient.  A call of
 * the optimizer (`optimize`, accel_rl/optimizers/single/ppo_optimizer.py:58-75) issues updates k = 0 .. n-1 and
 * then arl_opt_finish(n), which writes grad_norm_log[k % norm_log_len] for all of them and settles step_count.
 *   step_pp    f32[2] zero-initialised (Lasagne's t, ping-pong between consecutive updates)
 *   norm_parts f64[ARL_OPT_NORM_SLOTS][ARL_OPT_NORM_BLOCKS] scratch                                            */
#define ARL_OPT_NORM_SLOTS  64
#define ARL_OPT_NORM_BLOCKS 2048
int arl_opt_step_noclip(const arl_opt_state* opt, int32_t method, float learning_rate, float avg_factor,
                        float beta1_or_rho, float beta2, float epsilon, int32_t k, float* step_pp,
                        double* norm_parts, void* stream);
int arl_opt_finish(const arl_opt_state* opt, int32_t n_updates, float avg_factor, float* step_pp,
                   const double* norm_parts, void* stream);

/* The no-clip update in two parts, so that the bulk of it can leave the step's critical path: once the gradient of a
 * range [hole_first, hole_first + hole_count) of the bucket is final (spec 1: the first dense layer's 3.5 M weights,
 * written by its weight-gradient kernel long before the conv layers' backward ends), that range's update -- HBM-bound
 * streaming -- can run INSIDE the launch of a later MFMA-bound data-gradient kernel, in extra workgroups
 * (arl_corun_job below; inside the PPO step: the host launch 42.6 -> ~48 us, the step's own update launch
 * 19.9 -> 4.9 us), and the step ends with the update of the small rest.  part 0 = everything but the hole (advances t; hole_count = 0: the
 * plain arl_opt_step_noclip), part 1 = the hole as a launch of its own.  hole_first, hole_count multiples of 4.
 * Per element the arithmetic is arl_opt_step_noclip's; a call that used a hole ends with arl_opt_finish_split.       */
int arl_opt_step_noclip_split(const arl_opt_state* opt, int32_t method, float learning_rate, float avg_factor,
                              float beta1_or_rho, float beta2, float epsilon, int32_t k, float* step_pp,
                              double* norm_parts, int64_t hole_first, int64_t hole_count, int32_t part, void* stream);
int arl_opt_finish_split(const arl_opt_state* opt, int32_t n_updates, float avg_factor, float* step_pp,
                         const double* norm_parts, int64_t hole_count, void* stream);
/* Part 1 of update k as a job that a data-gradient launch carries: arl_corun_job_init describes it (same arguments
 * as part 1 above; nothing is launched), arl_conv2d_bwd_data / arl_conv2d_bwd_pair take it as an argument -- the
 * launch's grid gets one extra workgroup per CU (the first of the grid; ARL_CORUN_BLOCKS overrides the count, a tuning
 * aid) that streams the update while the others keep the matrix pipe busy -- and report whether they ran it;
 * arl_corun_job_run runs it as its own launch (what the caller does when no launch took it), before part 0.
 * The job is plain data owned by the caller: nothing is pending inside the library, an abandoned job costs nothing. */
typedef struct arl_corun_job { int64_t opaque[40]; } arl_corun_job;
int arl_corun_job_init(arl_corun_job* job, const arl_opt_state* opt, int32_t method, float learning_rate,
                       float avg_factor, float beta1_or_rho, float beta2, float epsilon, int32_t k, float* step_pp,
                       double* norm_parts, int64_t hole_first, int64_t hole_count);
int arl_corun_job_run(const arl_corun_job* job, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ACCEL_RL_HIP_H */
