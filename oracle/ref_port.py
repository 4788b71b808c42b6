"""
ORACLE / TEST INFRASTRUCTURE -- not part of the shipped product path.

CPU restatement (numpy) of the reference's algorithms on the hot path
(SURVEY.md section 8a).  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this file; the product (accel_rl_amd/) never does.

Parity status: PINNED.  Every function here is checked in
tests/test_oracle_golden.py against golden vectors produced by running the
reference's own code (/root/reference, imported with tests/golden/ref_shims.py)
in the build container -- see tests/golden/gen_golden.py.  Exceptions, which
stay "parity unpinned" because the arithmetic lives in absent third-party
packages: the ALE emulator (replaced by oracle/synth_ale.py per north_star) and
cv2.resize (restated as the rounded 2x2 box mean).

Numeric conventions.  The reference is 2018 numpy-1.x code; under numpy 2
(NEP 50) Python-float x float32 stays float32.  `promo="nep50"` reproduces the
reference bit-for-bit when it is run under numpy 2 (this container);
`promo="legacy"` reproduces the numpy-1.x promotion (Python float x float32
scalar -> float64).  The two differ by < 1e-5 (tests assert it).
"""

import numpy as np

from oracle import synth_ale

F32 = np.float32
F64 = np.float64

OBS_H, OBS_W = 104, 80           # accel_rl/envs/atari_env.py:13
CROP_ROWS = 2                    # accel_rl/envs/atari_env.py:155  (max_frame[:-2])


# =============================================================================
# Return / advantage scans          (accel_rl/algos/pg/util.py:6-37)
# =============================================================================

def gae_scan(rewards, values, dones, last_values, discount, gae_lambda,
             promo="nep50"):
    """GAE(lambda) over env-major [N,T] arrays.

    Restates gen_adv_est (accel_rl/algos/pg/util.py:6-23) for all N segments at
    once.  dtype walk of the reference expression (util.py:15-17):
      discount * vpred[t+1]      -> f32 (nep50) | f64 (legacy)
      ... * not_done[t] (int64)  -> f64
      rewards[t] + ... - vpred[t]-> f64
      lastgaelam                 -> f64 carry ; advantages[t] store rounds to f32
      returns = advantages + values  (f32 + f32, util.py:21)
    """
    r = np.asarray(rewards, F32)
    v = np.asarray(values, F32)
    d = np.asarray(dones).astype(bool)
    n_env, horizon = r.shape
    nd = (1 - d.astype(np.int64)).astype(F64)          # util.py:8
    adv = np.empty((n_env, horizon), F32)
    carry = np.zeros(n_env, F64)                       # util.py:13
    v_next = np.asarray(last_values, F32).reshape(n_env)
    g32 = F32(discount)
    gl = float(discount) * float(gae_lambda)           # util.py:17 (python floats)
    for t in range(horizon - 1, -1, -1):
        if promo == "nep50":
            gv = (g32 * v_next).astype(F64)
        else:
            gv = float(discount) * v_next.astype(F64)
        delta = (r[:, t].astype(F64) + gv * nd[:, t]) - v[:, t].astype(F64)
        carry = delta + (gl * nd[:, t]) * carry
        adv[:, t] = carry.astype(F32)
        v_next = v[:, t]
    ret = adv + v
    return adv, ret


def nstep_returns(rewards, dones, values, last_values, discount, promo="nep50"):
    """Discounted n-step return + advantage over env-major [N,T] arrays.

    Restates discount_returns (accel_rl/algos/pg/util.py:26-37) followed by
    `adv[:] = ret - v` (accel_rl/algos/pg/aac_base.py:121).
      nep50 : the running return stays np.float32 (`ret *= discount` is f32)
      legacy: `ret *= discount` promotes to float64; the store rounds to f32
    """
    r = np.asarray(rewards, F32)
    v = np.asarray(values, F32)
    d = np.asarray(dones).astype(bool)
    n_env, horizon = r.shape
    ret = np.empty((n_env, horizon), F32)
    if promo == "nep50":
        run = np.asarray(last_values, F32).reshape(n_env).copy()
        g = F32(discount)
        for t in range(horizon - 1, -1, -1):
            run = np.where(d[:, t], r[:, t], (run * g) + r[:, t]).astype(F32)
            ret[:, t] = run
    else:
        run = np.asarray(last_values, F32).reshape(n_env).astype(F64)
        g = float(discount)
        for t in range(horizon - 1, -1, -1):
            run = np.where(d[:, t], r[:, t].astype(F64), (run * g) + r[:, t].astype(F64))
            ret[:, t] = run.astype(F32)
    adv = ret - v
    return ret, adv


def valid_mask(reset_flags):
    """valids[e,t] = 1 for t <= first index where reset_flags[e,:] is set.

    Restates update_valids (accel_rl/algos/pg/util.py:56-63); `reset_flags` is
    env_infos.need_reset when present else dones (util.py:57).
    """
    f = np.asarray(reset_flags).astype(bool)
    n_env, horizon = f.shape
    first = np.where(f.any(axis=1), f.argmax(axis=1), horizon)   # horizon == never
    t = np.arange(horizon)[None, :]
    return (t <= first[:, None]).astype(np.int8)


def zero_invalid(valids, *arrays):
    """Restates zero_after_reset (util.py:40-46): zero everything past the mask.
    NB the reference zeroes agent_infos.value IN PLACE as well (util.py:46)."""
    out = []
    for a in arrays:
        a = np.array(a, copy=True)
        a[np.asarray(valids) == 0] = 0
        out.append(a)
    return out


def standardize(adv, valids=None):
    """(adv - mean) / (std + 1e-6), population std, optionally over valid
    samples only.  Restates accel_rl/algos/pg/aac_base.py:136-143."""
    a = np.array(adv, F32, copy=True)
    flat = a.reshape(-1)
    if valids is None:
        flat[:] = (flat - flat.mean()) / (flat.std() + 1e-6)
    else:
        idx = np.asarray(valids).reshape(-1).nonzero()
        sel = flat[idx]
        flat[idx] = (sel - sel.mean()) / (sel.std() + 1e-6)
    return a


def process_samples(rewards, dones, values, last_values, need_reset,
                    discount, gae_lambda, use_valids=False,
                    standardize_adv=False, promo="nep50"):
    """Restates AdvActorCriticBase.process_samples (aac_base.py:108-145) on
    env-major [N,T] arrays.  Returns dict(advantages, returns[, valids, value])."""
    if gae_lambda == 1:                                  # aac_base.py:115-121
        ret, adv = nstep_returns(rewards, dones, values, last_values, discount, promo)
    else:                                                # aac_base.py:122-127
        adv, ret = gae_scan(rewards, values, dones, last_values, discount,
                            gae_lambda, promo)
    out = dict()
    valids = None
    if use_valids:                                       # aac_base.py:129-134
        flags = dones if need_reset is None else need_reset
        valids = valid_mask(flags)
        adv, ret, val = zero_invalid(valids, adv, ret, values)
        out["valids"] = valids
        out["value"] = val
    if standardize_adv:                                  # aac_base.py:136-143
        adv = standardize(adv, valids)
    out["advantages"] = adv
    out["returns"] = ret
    return out


# =============================================================================
# Categorical action sampling       (rllab/misc/special.py:22-27)
# =============================================================================

def sample_actions(prob, uniforms):
    """k = #{j : cumsum_j(prob) < u}, clamped to A-1; fp32 sequential cumsum,
    compare in fp64.  Restates weighted_sample_n with the uniform variates made
    an explicit input (the reference draws np.random.rand(B), special.py:24)."""
    p = np.asarray(prob, F32)
    u = np.asarray(uniforms, F64).reshape(-1, 1)
    csum = np.cumsum(p, axis=1, dtype=F32)
    k = (csum.astype(F64) < u).sum(axis=1)
    n_act = p.shape[1]
    dtype = np.uint8 if n_act <= 256 else (np.uint16 if n_act <= 65536 else np.uint32)
    return np.minimum(k, n_act - 1).astype(dtype)       # spaces/discrete.py:14-19


# =============================================================================
# Frame preprocessing                (accel_rl/envs/atari_env.py:151-157)
# =============================================================================

def preprocess_pair(frame_a, frame_b, resample="box2x"):
    """max of two raw u8[210,160] frames, drop the last 2 rows, rounded 2x2 box
    mean -> u8[104,80].  `frame_a=None` means an all-zero first frame.
    resample="box2x": what atari_env.py:155 computes (cv2's default INTER_LINEAR at an exact 2x decimation -- the
    INTER_NEAREST constant sits in the `dst` slot); "nearest": what the call names, dst(y, x) = src(2y, 2x) (OpenCV's
    nearest source index is floor(dst * scale)).  Both restated from OpenCV's documented behaviour: cv2 is absent."""
    fb = np.asarray(frame_b, np.uint8).reshape(synth_ale.RAW_H, synth_ale.RAW_W)
    if frame_a is None:
        m = fb
    else:
        fa = np.asarray(frame_a, np.uint8).reshape(synth_ale.RAW_H, synth_ale.RAW_W)
        m = np.maximum(fa, fb)
    if resample == "nearest":
        return m[:synth_ale.RAW_H - CROP_ROWS][0::2, 0::2].copy()
    assert resample == "box2x", resample
    m = m[:synth_ale.RAW_H - CROP_ROWS].astype(np.uint16)
    box = m[0::2, 0::2] + m[0::2, 1::2] + m[1::2, 0::2] + m[1::2, 1::2]
    return ((box + 2) >> 2).astype(np.uint8)


# =============================================================================
# Environment wrapper                (accel_rl/envs/atari_env.py:16-191)
# =============================================================================

class PortedAtariEnv(object):
    """Restatement of AtariEnv over SynthALE.  `rng` stands for the constructing
    process's global numpy RNG (the reference calls np.random.randint,
    atari_env.py:97); pass a RandomState to emulate a worker process."""

    def __init__(self, game="pong", frame_skip=4, num_img_obs=4, clip_reward=True,
                 episodic_lives=True, max_start_noops=30,
                 repeat_action_probability=0., rng=None, pad_actions_to=None, resample="box2x"):
        self.rng = np.random if rng is None else rng
        self.game = game
        self.game_id, self.action_set, self.start_lives = synth_ale.GAMES[game]
        if pad_actions_to is not None:                   # suite mode (envs/synthetic_atari.py: padded_action_set)
            self.action_set = list(self.action_set) + [0] * (int(pad_actions_to) - len(self.action_set))
        self.bank = synth_ale.frame_bank(self.game_id)
        self.n_actions = len(self.action_set)
        self.frame_skip = frame_skip
        self.n_stack = num_img_obs
        self.clip_reward = clip_reward
        self.episodic_lives = episodic_lives
        self.max_start_noops = max_start_noops
        self.resample = resample
        self.has_fire = 1 in self.action_set             # atari_env.py:56 ("FIRE" = code 1)
        self.has_up = 2 in self.action_set               # atari_env.py:57 ("UP"   = code 2)
        # emulator state (SynthALE.loadROM draws the phase)
        self.phase = int(self.rng.randint(0, synth_ale.K_FRAMES))
        self.tick = 0
        self.emu_lives = self.start_lives
        self.over = False
        self.env_lives = 0
        self.stack = np.zeros((self.n_stack, OBS_H, OBS_W), np.uint8)
        self.first = None                                # raw_frame_1 (None == zeros)
        self.reset()                                     # atari_env.py:63

    # -- emulator (oracle/synth_ale.py) --------------------------------------
    def _emu_reset(self):
        self.tick = 0
        self.emu_lives = self.start_lives
        self.over = False

    def _emu_act(self, code):
        if self.over:
            return 0
        self.tick += 1
        r = synth_ale.reward_fn(self.tick, int(code))
        if self.start_lives > 0:
            if self.tick % synth_ale.LIFE_PERIOD == 0:
                self.emu_lives -= 1
                self.over = self.emu_lives == 0
        elif self.tick >= synth_ale.LIFE_PERIOD * 5:
            self.over = True
        return r

    def _screen(self):
        return self.bank[(self.phase + self.tick) % synth_ale.K_FRAMES]

    # -- wrapper --------------------------------------------------------------
    def _push_frame(self):
        """atari_env.py:151-157: grab frame 2, max with frame 1, resample, shift."""
        img = preprocess_pair(self.first, self._screen(), self.resample)
        self.stack = np.concatenate([self.stack[1:], img[None]])

    def _blank(self):
        """atari_env.py:159-163"""
        self.stack = np.zeros_like(self.stack)
        self.first = None

    def _press_start(self):
        """atari_env.py:172-179; rewards discarded"""
        self._emu_act(0)
        if self.has_fire:
            self._emu_act(1)
        if self.has_up:
            self._emu_act(2)
        self.env_lives = self.emu_lives

    def reset(self):
        """atari_env.py:93-100"""
        self._emu_reset()
        self._blank()
        self._press_start()
        for _ in range(int(self.rng.randint(0, self.max_start_noops + 1))):
            self._emu_act(0)
        self._push_frame()
        return self.stack.copy()

    def step(self, action):
        """atari_env.py:65-78 (+ :185-191 / :181-183 for the done rule)"""
        code = self.action_set[int(action)]
        reward = F32(0.)
        for _ in range(self.frame_skip - 1):
            reward = F32(reward + F32(self._emu_act(code)))
        self.first = self._screen().copy()               # _get_screen(1)
        reward = F32(reward + F32(self._emu_act(code)))
        self._push_frame()
        info = dict()
        if self.clip_reward:
            info["raw_reward"] = reward
            reward = F32(np.sign(reward))
        if self.episodic_lives:
            need_reset = bool(self.over)
            info["need_reset"] = need_reset
            lost = (self.emu_lives < self.env_lives) and (self.emu_lives > 0)
            if lost:
                self._press_start()
                self._blank()
                self._push_frame()
            done = bool(lost or need_reset)
        else:
            lost = (self.emu_lives < self.env_lives) and (self.emu_lives > 0)
            if lost:
                self._press_start()
            done = bool(self.over)
        return self.stack.copy(), reward, done, info


# =============================================================================
# Trajectory statistics              (accel_rl/sampler/util.py:75-101)
# =============================================================================

class PortedTrajInfo(dict):
    """Per-episode accumulators.  dtype walk under numpy 2: Return / RawReturn /
    DiscountedReturn accumulate np.float32 (python-number op f32 -> f32); the
    running discount is a python float (f64)."""

    def __init__(self, discount=1.):
        super().__init__(Length=0, Return=F32(0), RawReturn=F32(0),
                         NonzeroRewards=0, DiscountedReturn=F32(0))
        self.discount = 1. if discount is None else discount
        self.cur_discount = 1.

    def step(self, r, info):
        r = F32(r)
        self["Length"] += 1
        self["Return"] = F32(self["Return"] + r)
        self["RawReturn"] = F32(self["RawReturn"] + F32(info.get("raw_reward", r)))
        self["NonzeroRewards"] += int(r != 0)
        self["DiscountedReturn"] = F32(self["DiscountedReturn"] + F32(self.cur_discount) * r)
        self.cur_discount *= self.discount

    def as_tuple(self):
        return (int(self["Length"]), float(self["Return"]), float(self["RawReturn"]),
                int(self["NonzeroRewards"]), float(self["DiscountedReturn"]))


# =============================================================================
# The alternating act-server sampler, restated sequentially
#   (accel_rl/sampler/act_server/alternating/overlap/{sampler,worker}.py)
# =============================================================================

class CpuSamplerPort(object):
    """Single-process restatement of ActsrvAltOvrlpSampler + its 2*n_parallel
    worker processes.  Worker `w` (group-major order, sampler.py:165-185) owns
    envs [w*envs_per, (w+1)*envs_per) and the RNG stream RandomState(seed + w)
    (the reference seeds each worker process's global numpy RNG with seed + i,
    sampler/util.py:68-69).  Results do not depend on process timing in the
    reference (every hand-off is semaphore-ordered), so a sequential walk in
    (step, group, worker, env) order reproduces them exactly."""

    def __init__(self, game, horizon, n_parallel=1, envs_per=1,
                 max_path_length=np.inf, mid_batch_reset=True, env_kwargs=None,
                 eval_steps=None, eval_envs_per=None):
        """eval_envs_per / eval_steps: the AAOEvalSampler variant (sampler_with_eval.py:6-54,
        worker_with_eval.py): extra evaluation envs per worker, and the over-length rule of
        BOTH collectors becomes Length >= max_path_length (worker_with_eval.py:48,123,159)
        where the plain worker has > (worker.py:42)."""
        self.eval_envs_per = eval_envs_per
        if eval_envs_per is not None:
            self.n_eval_envs = eval_envs_per * n_parallel * 2
            self.eval_horizon = eval_steps // self.n_eval_envs
        self.game = game
        self.horizon = horizon
        self.n_parallel = n_parallel
        self.envs_per = envs_per
        self.max_path_length = max_path_length
        self.mid_batch_reset = mid_batch_reset
        self.env_kwargs = dict(env_kwargs or {})
        self.n_envs = 2 * n_parallel * envs_per
        self.half = n_parallel * envs_per

    def initialize(self, seed, discount=1., master_rng=None):
        """sampler.py:40-79 + worker start-up (worker.py:116-141, util.py:26-57
        with max_decorrelation_steps == 0)."""
        n, t = self.n_envs, self.horizon
        self.discount = discount
        # master: example env + build_env_buffer draws (act_server/buffers.py:7-12)
        mrng = np.random if master_rng is None else master_rng
        example = PortedAtariEnv(game=self.game, rng=mrng, **self.env_kwargs)
        example.reset()
        example.step(int(mrng.randint(example.n_actions, dtype=np.uint8)))
        self.n_actions = example.n_actions
        f = example.n_stack
        self.buf = dict(
            observations=np.zeros((n * t, f, OBS_H, OBS_W), np.uint8),
            rewards=np.zeros(n * t, F32),
            dones=np.zeros(n * t, bool),
            raw_reward=np.zeros(n * t, F32),
            need_reset=np.zeros(n * t, bool),
            actions=np.zeros(n * t, np.uint8),
            prob=np.zeros((n * t, self.n_actions), F32),
            value=np.zeros(n * t, F32),
            extra_observations=np.zeros((n, f, OBS_H, OBS_W), np.uint8),
        )
        self.step_obs = np.zeros((n, f, OBS_H, OBS_W), np.uint8)
        self.envs, self.trajs, self.eval_envs = [], [], []
        for w in range(2 * self.n_parallel):
            rng = np.random.RandomState((seed + w) % 4294967294)   # ext.set_seed
            for _ in range(self.envs_per):
                env = PortedAtariEnv(game=self.game, rng=rng, **self.env_kwargs)
                self.envs.append(env)
            for _ in range(self.eval_envs_per or 0):                # worker_with_eval.py:200
                self.eval_envs.append(PortedAtariEnv(game=self.game, rng=rng, **self.env_kwargs))
            for i in range(self.envs_per):                          # start_envs
                e = w * self.envs_per + i
                self.step_obs[e] = self.envs[e].reset()
                self.trajs.append(PortedTrajInfo(discount))
        self.frozen = [False] * n
        self.reset_flags = np.zeros(n, bool)             # step_bufs[g].reset (act_server/buffers.py:24-30)
        return self.n_actions, n * t

    def obtain_samples(self, policy):
        """One batch.  `policy.get_actions(obs[B]) -> (acts, dict(prob, value))`
        is called once per (step, group) with the group's B = N/2 observations,
        in the reference's order (sampler.py:129-145)."""
        n, t, half = self.n_envs, self.horizon, self.half
        b = self.buf
        completed = []
        for e in range(n):                                          # worker.py:30-32
            b["observations"][e * t] = self.step_obs[e]
        frozen = [False] * n
        for s in range(t):
            for j in (0, 1):
                lo, hi = j * half, (j + 1) * half
                if self.reset_flags[lo:hi].any():                   # sampler.py:135-138 (for recurrence)
                    for i in np.where(self.reset_flags[lo:hi])[0]:
                        if hasattr(policy, "reset_one"):            # BasePolicy.reset_one is a no-op
                            policy.reset_one(idx=i)
                    self.reset_flags[lo:hi] = False
                acts, infos = policy.get_actions(self.step_obs[lo:hi])
                idx = np.arange(lo, hi) * t + s                     # sampler.py:143-145
                b["actions"][idx] = acts
                for k, v in infos.items():                          # prob, value (+ previous hidden states)
                    if k not in b:
                        b[k] = np.zeros((n * t,) + np.asarray(v).shape[1:], np.asarray(v).dtype)
                    b[k][idx] = v
                for e in range(lo, hi):
                    if not self.mid_batch_reset and frozen[e]:      # worker.py:80
                        continue
                    env, traj = self.envs[e], self.trajs[e]
                    o, r, d, info = env.step(acts[e - lo])
                    traj.step(r, info)
                    over_len = (traj["Length"] >= self.max_path_length) if self.eval_envs_per is not None \
                        else (traj["Length"] > self.max_path_length)
                    hit = over_len or (d and info.get("need_reset", True))
                    if hit:                                         # worker.py:42-50
                        d = True
                        if self.eval_envs_per is None:              # worker.py:46,88 (worker_with_eval.py omits it)
                            self.reset_flags[e] = True
                        if over_len and "need_reset" in info:
                            info["need_reset"] = True
                        completed.append(traj)
                        self.trajs[e] = PortedTrajInfo(self.discount)
                        if self.mid_batch_reset:
                            o = env.reset()
                        else:
                            frozen[e] = True                        # worker.py:88-95
                    if self.mid_batch_reset or not hit:
                        self.step_obs[e] = o
                        if s < t - 1:
                            b["observations"][e * t + s + 1] = o
                    b["rewards"][e * t + s] = r
                    b["dones"][e * t + s] = d
                    if "raw_reward" in info:
                        b["raw_reward"][e * t + s] = info["raw_reward"]
                    if "need_reset" in info:
                        b["need_reset"][e * t + s] = info["need_reset"]
        b["extra_observations"][:] = self.step_obs                  # sampler.py:147-151
        if not self.mid_batch_reset:                                # worker.py:108-113
            for e in range(n):
                if frozen[e]:
                    self.step_obs[e] = self.envs[e].reset()
        return b, completed


    def evaluate_policy(self, policy):
        """AAOEvalSampler.evaluate_policy + collect_eval (sampler_with_eval.py:20-54,
        worker_with_eval.py:148-176): fresh resets of the evaluation envs, eval_horizon served
        steps, nothing stored; returns the completed TrajInfos (fresh ones per call)."""
        ne, half = self.n_eval_envs, self.n_eval_envs // 2
        obs = np.stack([env.reset() for env in self.eval_envs])
        trajs = [PortedTrajInfo(self.discount) for _ in range(ne)]
        completed = []
        for _ in range(self.eval_horizon):
            for j in (0, 1):
                lo, hi = j * half, (j + 1) * half
                acts, _ = policy.get_actions(obs[lo:hi])
                for e in range(lo, hi):
                    env = self.eval_envs[e]
                    o, r, d, info = env.step(acts[e - lo])
                    trajs[e].step(r, info)
                    if trajs[e]["Length"] >= self.max_path_length or (d and info.get("need_reset", True)):
                        o = env.reset()
                        completed.append(trajs[e])
                        trajs[e] = PortedTrajInfo(self.discount)
                    obs[e] = o
        return completed


# =============================================================================
# Optimiser-side host logic
# =============================================================================

def minibatch_indices(batch_size, data_length, shuffle, rng=None):
    """Restates iterate_mb_idxs (accel_rl/optimizers/util.py:8-18): one fresh
    permutation per call, tail dropped."""
    rng = np.random if rng is None else rng
    if shuffle:
        perm = np.arange(data_length)
        rng.shuffle(perm)
    out = []
    for start in range(0, data_length - batch_size + 1, batch_size):
        if shuffle:
            out.append(perm[start:start + batch_size])
        else:
            out.append(np.arange(start, start + batch_size))
    return out


def n_itr_for(n_steps, sample_size, log_steps):
    """Restates AccelRLBase.get_n_itr (accel_rl/runners/accel_rl_base.py:74-87)."""
    log_itrs = max(log_steps // sample_size, 1)
    n_itr = max(n_steps // sample_size, 1)
    rem = n_itr % log_itrs
    n_itr = n_itr - rem if rem <= log_itrs / 2. else n_itr + (log_itrs - rem)
    return n_itr + 1, log_itrs


def rmsprop_step(p, g, acc, lr, rho=0.9, eps=1e-6):
    """accel_rl/optimizers/update_methods_stats.py:11-33 (Lasagne rmsprop), f32."""
    p, g, acc = (np.asarray(x, F32) for x in (p, g, acc))
    acc_new = F32(rho) * acc + (F32(1) - F32(rho)) * g * g
    step = F32(lr) * g / np.sqrt(acc_new + F32(eps))
    return (p - step).astype(F32), acc_new.astype(F32)


def adam_step(p, g, m, v, t_prev, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """accel_rl/optimizers/update_methods_stats.py:55-87 (Lasagne adam), f32.
    t is the f32 step counter AFTER increment; a_t = lr*sqrt(1-b2^t)/(1-b1^t)."""
    p, g, m, v = (np.asarray(x, F32) for x in (p, g, m, v))
    t = F32(t_prev) + F32(1)
    a_t = F32(lr) * np.sqrt(F32(1) - F32(beta2) ** t) / (F32(1) - F32(beta1) ** t)
    m_t = F32(beta1) * m + (F32(1) - F32(beta1)) * g
    v_t = F32(beta2) * v + (F32(1) - F32(beta2)) * g * g
    step = F32(a_t) * m_t / (np.sqrt(v_t) + F32(eps))
    return (p - step).astype(F32), m_t.astype(F32), v_t.astype(F32), t


def clip_by_total_norm(grads_flat, clip):
    """Lasagne total_norm_constraint as used by apply_grad_norm_clip
    (accel_rl/optimizers/util.py:70-76): norm over ALL params;
    g *= clip(norm, 0, c) / (1e-7 + norm); clip None -> grads untouched."""
    g = np.asarray(grads_flat, F32)
    norm = np.sqrt(np.sum(g.astype(F32) ** 2, dtype=F32))
    if clip is None:
        return g, norm
    scale = np.clip(norm, 0, F32(clip)) / (F32(1e-7) + norm)
    return (g * F32(scale)).astype(F32), norm


# =============================================================================
# PPO surrogate: value and gradient with respect to the likelihood ratio
#   (accel_rl/algos/pg/ppo.py:42-51)
# =============================================================================

def ppo_surrogate(ratio, adv, clip, tie_rule="theano"):
    """surr = minimum(ratio adv, clip(ratio, 1 - clip, 1 + clip) adv) (ppo.py:45-49) and d surr / d ratio, f32.

    PARITY UNPINNED for the gradient: it is produced by Theano's symbolic differentiation, and Theano (unpinned
    in the reference's environment.yml, absent from /root/reference and from this image) cannot be run here.
    Restated from its published scalar-op rules, theano/scalar/basic.py.  The reference imports theano.gpuarray
    (accel_rl/runners/accel_rl_base.py:62-64) and names "Theano 0.9" / "Theano 1.0" in its comments
    (accel_rl/algos/dqn/cat_dqn.py:85-86), i.e. it runs on Theano >= 0.9, where
        Minimum.L_op:  e = eq(minimum(x, y), x);  gx = e gz;  gy = (1 - e) gz
                       ("This form handle the case when both value are the same. In that case, gx will be gz, gy
                        will be 0."; theano/tensor/tests/test_basic.py::test_maximum_minimum_grad: at x == y the
                        gradients are [[1], [0]] -- "we only pass the gradient to the first input in that case")
        Clip.L_op:     gx = ((x >= min) & (x <= max)) gz                              (bounds included)
    so with s1 = ratio adv (the FIRST argument, ppo.py:49), s2 = clip(ratio) adv:
        tie_rule="theano":  d surr / d ratio = adv [surr == s1] + adv [surr != s1] [lo <= ratio <= hi]
    = adv inside the clip range (the tie goes to the unclipped branch alone), adv where s1 < s2 outside it, 0 where the
    clipped branch alone is the minimum.  tie_rule="math" is the mathematical derivative (equal to the above except
    where s1 == s2 by rounding outside the range); tie_rule="both" is Theano <= 0.7 (gx = eq(min, x) gz AND gy =
    eq(min, y) gz): 2 adv inside the range."""
    ratio, adv = np.asarray(ratio, F32), np.asarray(adv, F32)
    lo, hi = F32(1) - F32(clip), F32(1) + F32(clip)
    s1 = ratio * adv
    s2 = np.minimum(np.maximum(ratio, lo), hi) * adv
    surr = np.minimum(s1, s2)
    inside = (ratio >= lo) & (ratio <= hi)
    first = surr == s1
    if tie_rule == "theano":
        grad = adv * first.astype(F32) + adv * (~first & inside).astype(F32)
    elif tie_rule == "both":
        grad = adv * first.astype(F32) + adv * ((surr == s2) & inside).astype(F32)
    elif tie_rule == "math":
        grad = np.where(inside, adv, np.where(s1 < s2, adv, F32(0))).astype(F32)
    else:
        raise ValueError(tie_rule)
    return surr.astype(F32), grad.astype(F32)
