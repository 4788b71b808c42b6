"""GpuVecSampler: the whole rollout on one MI355X.

Drop-in for the reference's ActsrvAltOvrlpSampler
(accel_rl/sampler/act_server/alternating/overlap/sampler.py:20-151 plus its
worker processes, overlap/worker.py:116-153): same constructor, same
initialize / policy_init / obtain_samples / shutdown contract, same
`samples_buf` keys, dtypes and env-major layout -- but the 2*n_parallel CPU
worker processes, their semaphores and the per-step host<->device copies are
gone.  Per agent step the device runs: policy forward (torch) -> act_step kernel
(sample action, emulate, bookkeeping) -> frame_step kernel (pixels, stores).
The whole horizon can be captured once in a hipGraph and replayed per batch.

What stays on the host, to keep results identical to the reference on the same
seeds (DESIGN.md "RNG streams"): numpy's global MT19937 supplies the action
uniforms (np.random.rand, rllab/misc/special.py:24) -- drawn for the whole batch
in one call and uploaded once -- and one RandomState per *simulated* worker
(seed + i, sampler.py:179) supplies emulator phases and start no-ops, pre-drawn
into a device ring that env.hip consumes in the reference's order.

`n_parallel` / `envs_per` keep their meaning for layout and RNG streams only:
env index e = (group * n_parallel + rank) * envs_per + i (sampler.py:160-185).
"""
import ctypes
import os

import numpy as np
import torch

from accel_rl_amd import _lib
from accel_rl_amd.buffers import (buffer_with_segs_view, batch_buffer, combine_distinct_buffers,
                                  buffer_length, count_buffer_size)
from accel_rl_amd.envs import synthetic_atari as synth
from accel_rl_amd.sampler.base import BaseMbSampler
from accel_rl_amd.sampler.util import TrajInfo
from accel_rl_amd.util import logger
from accel_rl_amd.util.misc import graph_capture_mode, nbytes_unit, struct

NOOP_RING = 4096


def _packed_block(spec, device, pinned=False):
    """One zeroed byte block holding every (name, shape, dtype) of `spec` at 128-byte-aligned offsets (the arrival
    tickets of arl_env_step sit on a cache line each only if `epoch` starts on one); returns (block, {name: typed view})."""
    offsets, total = [], 0
    for _, shape, dtype in spec:
        total = (total + 127) // 128 * 128
        offsets.append(total)
        total += int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
    block = torch.zeros(total, dtype=torch.uint8, device=device)
    if pinned:
        block = block.pin_memory()
    views = dict()
    for (name, shape, dtype), off in zip(spec, offsets):
        nbytes = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        views[name] = block[off:off + nbytes].view(dtype).view(*shape)
    return block, views


class _LazyTrajInfos(list):
    """The batch's completed TrajInfos, fetched on first use.  obtain_samples returns right after
    enqueuing the rollout; a caller that first enqueues the learner (as the runners do) and only then
    looks at the trajectory statistics waits for the ROLLOUT's event while the learner already runs."""

    def __init__(self, sampler, host_set=None, older=None):
        super().__init__()
        self._sampler, self._set, self._older = sampler, host_set, older

    def resolve(self):
        smp, self._sampler = self._sampler, None
        if smp is not None:
            older, self._older = self._older, None
            if older is not None:                    # batches are read out in order (the no-op ring's top-up counts on it)
                older.resolve()
            if self._set is not None and self._set.pending is self:
                self._set.pending = None
            super().extend(smp._drain_traj_infos(self._set))
        return self

    def __len__(self):
        self.resolve()
        return super().__len__()

    def __iter__(self):
        self.resolve()
        return super().__iter__()

    def __getitem__(self, i):
        self.resolve()
        return super().__getitem__(i)

    def __bool__(self):
        return len(self) > 0

    def __eq__(self, other):
        self.resolve()
        return super().__eq__(other)

    def __repr__(self):
        self.resolve()
        return super().__repr__()


class GpuVecSampler(BaseMbSampler):

    _alias_extra_obs = True     # (A/B switch of policy_init's aliasing of extra_observations onto step_obs)
    # one launch per step for everything per-env (arl_env_step_served): the policy's last fold + output layers, the action
    # draw, the env step and the next observation's first convolution (A/B switch; bit-identical either way)
    _serve_in_step = os.environ.get("ARL_SERVE_IN_STEP", "1") != "0"
    # ... while one 16-wave workgroup per env (one per CU at a time) beats the separate launches: measured up to 1024 envs
    # when conv 1 rides along (spec 1: -15 % at 256, -4 % at 1024; spec 0: -23 % / -5 %), up to 512 when it does not (a first
    # layer the launch does not take: -13 % at 256, -4 % at 512, +1 % at 1024; tools/serve_step_probe.py,
    # profiles/r06/serve_step_probe*.txt)
    _serve_in_step_max_envs = (512, 1024)       # (without, with conv 1 in the launch)
    _two_host_sets = os.environ.get("ARL_TWO_HOST_SETS", "1") != "0"     # (A/B switch, see _sets)

    def __init__(self, n_parallel=1, envs_per=1, device=None, use_graph=True, **kwargs):
        super().__init__(n_parallel=n_parallel, envs_per=envs_per, **kwargs)
        self._total_n_envs = 2 * n_parallel * envs_per
        self.device = device
        self.use_graph = use_graph
        self._sets, self._batch_no = [], 0

    # Host-side per-batch buffers (pinned uniforms in, pinned results out, event, captured graph, lazily read trajectory
    # records) come in TWO sets when batches are graph replays: batch i + 1 can be enqueued while batch i still runs, its
    # results unread -- a caller that only samples (evaluation, data collection) then keeps the device busy back to back
    # instead of paying a host round trip per batch.  samples_buf itself is one buffer, as the reference's: consume it
    # before the next call.
    @property
    def _graph(self):
        """The captured rollout (of the first buffer set); assigning None drops every set's graph."""
        return self._sets[0].graph if self._sets else None

    @_graph.setter
    def _graph(self, value):
        for h in self._sets:
            h.graph = value

    @property
    def _pending(self):
        """A batch whose trajectory records have not been read yet, if any; assigning None forgets them all."""
        return next((h.pending for h in self._sets if h.pending is not None), None)

    @_pending.setter
    def _pending(self, value):
        for h in self._sets:
            h.pending = value
        if value is None:
            self._last_pending = None

    # ------------------------------------------------------------------ API
    def initialize(self, seed, affinities=None, discount=1, need_extra_obs=False):
        if not getattr(self.EnvCls, "batched_device_env", False):
            raise TypeError("GpuVecSampler needs an EnvCls with the batched device protocol "
                            "(e.g. SynthAtariEnv); got {}".format(self.EnvCls))
        _lib.load()                                       # fail loudly, before anything else
        affinities = affinities or dict()
        if self.device is None:
            self.device = torch.device("cuda", int(affinities.get("gpu", 0) or 0))
        self.device = torch.device(self.device)
        dev = self.device
        n, t = self._total_n_envs, self.horizon
        self.seed = seed
        self.discount = 1. if discount is None else discount
        self.need_extra_obs = need_extra_obs
        self.sample_size = n * t

        # -- master-side example env + example transition: same global-RNG draws
        #    as sampler.py:42 and act_server/buffers.py:8-9
        env = self.EnvCls(**self.env_args)
        synth.draw_noops(np.random, env.max_start_noops)           # env.reset()
        env.action_space.sample()                                  # env.step(sample())
        self.env = env
        self.env_spec = env.spec
        n_act = env.action_space.n
        f = env.num_img_obs
        obs_example = torch.zeros((f, synth.OBS_H, synth.OBS_W), dtype=torch.uint8)
        env_infos = dict()
        if env.clip_reward:
            env_infos["raw_reward"] = np.float32(0)
        if env.episodic_lives:
            env_infos["need_reset"] = False
        examples = dict(observations=obs_example, rewards=np.float32(0), dones=False,
                        env_infos=env_infos)
        self.envs_buf = buffer_with_segs_view(examples, n * t, t, dev)
        if need_extra_obs:
            self.envs_buf.extra_observations = batch_buffer(obs_example, n, dev)
        # step buffers: observation_space.sample() + action_space.sample() per group
        # (act_server/buffers.py:24-30) -- drawn only to keep the RNG stream aligned
        eval_per = getattr(self, "eval_envs_per", None)            # set by GpuVecEvalSampler
        for _ in range(4 if eval_per is not None else 2):          # (+ 2 eval step buffers, sampler.py:204-206)
            env.observation_space.sample()
            env.action_space.sample()
        self.step_obs = torch.zeros((n, f, synth.OBS_H, synth.OBS_W), dtype=torch.uint8, device=dev)

        # -- game description + frame bank in HBM
        self.bank = torch.from_numpy(synth.frame_bank(env.game_id)).to(dev)
        g = _lib.ArlGame()
        g.bank = self.bank.data_ptr()
        g.n_frames, g.n_actions = synth.K_FRAMES, n_act
        for i, code in enumerate(env.action_set):
            g.action_set[i] = code
        g.start_lives, g.life_period = env.start_lives, synth.LIFE_PERIOD
        g.frame_skip, g.n_stack = env.frame_skip, f
        g.clip_reward, g.episodic_lives = int(env.clip_reward), int(env.episodic_lives)
        g.resample_mode = _lib.RESAMPLE_MODES[getattr(env, "resample", "box2x")]
        self._game = g

        # -- per-env state (SoA) and the simulated workers' RNG streams
        n_streams = 2 * self.n_parallel
        i32 = lambda *s: torch.zeros(s, dtype=torch.int32, device=dev)      # noqa: E731
        u8 = lambda *s: torch.zeros(s, dtype=torch.uint8, device=dev)       # noqa: E731
        f32 = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)    # noqa: E731
        self._st = struct(
            tick=i32(n), emu_lives=i32(n), env_lives=i32(n), phase=i32(n), over=u8(n),
            frozen=u8(n), traj_len=i32(n), traj_nonzero=i32(n), traj_ret=f32(n), traj_raw=f32(n),
            traj_disc=f32(n), traj_curdisc=torch.ones(n, dtype=torch.float64, device=dev),
            frame_a=i32(n), frame_b=i32(n), frame_mode=u8(n), reset_flag=u8(n),
            noop_ring=u8(n_streams, NOOP_RING), next_reset=u8(2, n), launch_count=i32(1),
        )
        # the batch's small results live in ONE block so that one D2H copy mirrors them (see _host)
        spec = (("noop_cursor", (2, n_streams), torch.int64), ("epoch", (_lib.EPOCH_WORDS,), torch.int32),
                ("done_count", (1,), torch.int32), ("done_int", (n * t, 3), torch.int32),
                ("done_flt", (n * t, 3), torch.float32))
        self._results_block, views = _packed_block(spec, dev)
        self._st.update(views)
        self._results_spec = spec
        self._worker_rngs = []
        self._eval_phases = np.zeros(n_streams * (eval_per or 0), np.int32)
        phases = np.zeros(n, np.int32)
        ring = np.zeros((n_streams, NOOP_RING), np.uint8)
        for w in range(n_streams):                       # group-major worker order
            rs = np.random.RandomState((seed + w) % 4294967294)   # initialize_worker -> set_seed
            for i in range(self.envs_per):               # envs = [EnvCls(...) ...] (worker.py:122)
                phases[w * self.envs_per + i] = synth.draw_phase(rs)
                synth.draw_noops(rs, env.max_start_noops)
            for i in range(eval_per or 0):               # eval_envs = [...] (worker_with_eval.py:200)
                self._eval_phases[w * eval_per + i] = synth.draw_phase(rs)
                synth.draw_noops(rs, env.max_start_noops)
            if env.max_start_noops > 0:
                ring[w] = synth.draw_noops(rs, env.max_start_noops, size=NOOP_RING)
            self._worker_rngs.append(rs)
        self._ring_host = ring
        self._ring_produced = np.full(n_streams, NOOP_RING, np.int64)
        self._st.phase.copy_(torch.from_numpy(phases))
        self._st.noop_ring.copy_(torch.from_numpy(ring))

        s = _lib.ArlEnvState()
        s.n_env = n
        for k in ("tick", "emu_lives", "env_lives", "phase", "over", "frozen", "traj_len",
                  "traj_nonzero", "traj_ret", "traj_raw", "traj_disc", "traj_curdisc", "frame_a",
                  "frame_b", "frame_mode", "reset_flag", "noop_ring", "noop_cursor", "epoch",
                  "done_count", "done_int", "done_flt", "next_reset", "launch_count"):
            setattr(s, k, self._st[k].data_ptr())
        s.noop_ring_len, s.envs_per_stream, s.done_capacity = NOOP_RING, self.envs_per, n * t
        self._state = s

        # -- start_envs (sampler/util.py:26-33): reset every env, in env order per stream
        self._rollout = None
        ro = self._make_rollout(None)
        with torch.cuda.device(dev):
            _lib.env_reset(self._game, self._state, ro, None, env.max_start_noops)
            if self.max_decorrelation_steps:
                self._decorrelate()
        return self.env_spec, self.sample_size, self.horizon, self.mid_batch_reset

    def policy_init(self, policy):
        """reference: sampler.py:81-95"""
        dev, n, t = self.device, self._total_n_envs, self.horizon
        self.policy = policy
        n_act = self.env_spec.action_space.n
        # build_policy_buffer (act_server/buffers.py:33-38): example action + agent_info;
        # the reference evaluates the policy once on a random observation
        policy.reset(n_batch=1)
        example_obs = self.env_spec.observation_space.sample()
        policy.get_action(torch.from_numpy(example_obs).to(dev))
        agent_infos = dict(prob=np.zeros(n_act, np.float32), value=np.float32(0))
        self._recurrent = bool(getattr(policy, "recurrent", False))
        if self._recurrent:                               # previous hidden state of every step (base.py:86-93)
            if not hasattr(policy, "act_step"):
                raise NotImplementedError("recurrent policies must provide act_step / reset_rows")
            for key, state in zip(policy.state_info_keys, policy.get_prev_hiddens()):
                agent_infos[key] = np.zeros(state.shape[1], np.float32)
            self._prev_frozen = torch.zeros(n, dtype=torch.uint8, device=dev)
        examples = dict(actions=np.zeros((), self.env_spec.action_space.dtype), agent_infos=agent_infos)
        policy_buf = buffer_with_segs_view(examples, n * t, t, dev)
        self.samples_buf = combine_distinct_buffers(self.envs_buf, policy_buf)
        assert buffer_length(self.samples_buf) == self.sample_size
        policy.reset(n_batch=n if self._recurrent else self.n_parallel * self.envs_per)
        self._rollout = self._make_rollout(self.samples_buf)
        # The current observation of env e at step s is row e * t + s of the rollout buffer.  A policy that serves
        # rows of a buffer in place lets the step kernel write each stacked observation once (no second,
        # contiguous copy kept current in step_obs); needs every env to step at every step (mid_batch_reset).
        self._single_write = bool(self.mid_batch_reset and not self._recurrent and
                                  getattr(policy, "serves_rows", False) and self._kernel_max_path_length() >= 1)
        # With mid-batch resets nothing touches step_obs between the batch's last step and the next batch's first, and at
        # the end of a batch it IS the bootstrap observation of every env (sampler.py:147-151 copies step_buf.obs): the
        # rollout buffer's extra_observations is then the same memory, and the copy at the end of every batch (5.7 us of a
        # 515 us rollout at 256 envs) is not made.  Without mid-batch resets frozen envs are reset AFTER the reference has
        # copied (worker.py:108-113), so the copy stays.
        self._serve_fused = bool(self._single_write and type(self)._serve_in_step and self._game.n_stack <= 4 and
                                 hasattr(policy, "serve_forward") and policy.serve_supported())
        if self._serve_fused:
            with_conv1 = policy.serve_conv1(self._game, n)[0] is not None
            self._serve_fused = n <= type(self)._serve_in_step_max_envs[int(with_conv1)]
        self._extra_is_step_obs = bool(self.need_extra_obs and self.mid_batch_reset and type(self)._alias_extra_obs)
        if self._extra_is_step_obs:
            self.samples_buf.extra_observations = self.envs_buf.extra_observations = self.step_obs
        self._step_rows = (torch.arange(n, dtype=torch.int32, device=dev)[None, :] * t +
                           torch.arange(t, dtype=torch.int32, device=dev)[:, None]).contiguous()
        self._uniforms = torch.empty((t, n), dtype=torch.float64, device=dev)
        # pinned mirrors of the batch's small results (completed-episode records, no-op ring cursors):
        # copied at the end of the batch ON the stream (inside the hipGraph), read by the host after
        # waiting for the batch event only -- not for whatever was enqueued behind it (the learner)
        self._sets, self._batch_no, self._last_pending = [], 0, None
        # (two sets only where the HOST prepares nothing else per batch: a policy with its own per-batch draws -- the DQN
        #  family's epsilon-greedy override table, staged through one pinned buffer -- or recurrent state keeps the strict
        #  hand-over, in which a batch's host-side preparation starts when the previous batch has finished)
        overlap = (self.use_graph and type(self)._two_host_sets and not hasattr(policy, "host_draws") and not self._recurrent)
        for _ in range(2 if overlap else 1):
            block, views = _packed_block(self._results_spec, "cpu", pinned=True)
            self._sets.append(struct(uniforms_host=torch.empty(t * n, dtype=torch.float64).pin_memory(), host_block=block,
                                     host=struct(**views), event=torch.cuda.Event(), pending=None, graph=None))
        logger.log("GpuVecSampler -- total_n_envs: {}".format(self.total_n_envs))
        logger.log("GpuVecSampler -- batch buffer size: {:,.1f} {}".format(
            *nbytes_unit(count_buffer_size(self.samples_buf))))

    def obtain_samples(self, itr):
        """reference: sampler.py:97-104 (+ serve_actions :120-151)"""
        n, t = self._total_n_envs, self.horizon
        cur = self._sets[self._batch_no % len(self._sets)]
        self._batch_no += 1
        if cur.pending is not None:                      # the records of the batch that used these pinned mirrors last must
            cur.pending.resolve()                        # be read out before this batch overwrites them
        # one np.random.rand(B) per (step, group) in the reference == one flat draw here; a policy
        # with its own action randomness (epsilon-greedy) makes the reference's draws itself
        draws = self.policy.host_draws(t, n) if hasattr(self.policy, "host_draws") else np.random.rand(t * n)
        cur.uniforms_host.copy_(torch.from_numpy(draws))
        with torch.cuda.device(self.device):
            if self.use_graph:
                if cur.graph is None:
                    self._capture(cur)
                cur.graph.replay()
            else:
                self._enqueue_batch(cur)
            cur.event.record()
        older = self._last_pending if (self._last_pending is not None and self._last_pending._sampler is not None) else None
        cur.pending = self._last_pending = _LazyTrajInfos(self, cur, older)
        return self.samples_buf, cur.pending

    def shutdown(self):
        for h in self._sets:
            h.graph = None

    @property
    def alternating(self):
        # every env is served in one forward per step; nothing alternates on the device
        return False

    # -------------------------------------------------------------- internals
    def _make_rollout(self, buf):
        ro = _lib.ArlRollout()
        ro.horizon = self.horizon
        ro.step_obs = self.step_obs.data_ptr()
        if buf is not None:
            ro.observations = buf.observations.data_ptr()
            ro.rewards, ro.dones = buf.rewards.data_ptr(), buf.dones.data_ptr()
            ro.raw_reward = buf.env_infos["raw_reward"].data_ptr() if "raw_reward" in buf.env_infos else None
            ro.need_reset = buf.env_infos["need_reset"].data_ptr() if "need_reset" in buf.env_infos else None
            ro.actions = buf.actions.data_ptr()
            ro.prob, ro.value = buf.agent_infos["prob"].data_ptr(), buf.agent_infos["value"].data_ptr()
        return ro

    def _enqueue_batch(self, cur):
        """All device work of one batch, on the current stream (graph-capturable); cur: the host-side buffer set."""
        n, t = self._total_n_envs, self.horizon
        buf, ro, env = self.samples_buf, self._rollout, self.env
        _lib.copy_bytes(self._uniforms, cur.uniforms_host)     # (a kernel node reading the pinned buffer: no memcpy node)
        # observations[:, 0] = step_obs (worker.py:30-32), done_count = 0 -- and, when the steps are served in one launch
        # each, conv 1 of those rows from the same pass over them (y1: conv 1 of the step's observations, once a launch
        # has left it)
        conv1, y1 = self.policy.serve_conv1(self._game, n) if self._serve_fused else (None, None)
        if conv1 is not None:
            _lib.rollout_begin_conv1(self._game, self._state, ro, conv1)
        else:
            _lib.rollout_begin(self._game, self._state, ro)
        for s in range(t):
            if hasattr(self.policy, "set_step"):
                self.policy.set_step(s)
            if self._serve_fused:
                head, conv1, y1 = self.policy.serve_forward(self._game, buf.observations, self._step_rows[s], y1,
                                                            want_next=s + 1 < t)
                _lib.env_step_served(self._game, self._state, ro, head, conv1, self._uniforms[s], s,
                                     self._kernel_max_path_length(), self.discount, env.max_start_noops)
                continue
            if self._recurrent:
                prob, value, *prev = self.policy.act_step(self.step_obs)
                for key, state in zip(self.policy.state_info_keys, prev):      # stored at (env, step)
                    buf.agent_infos[key].view(n, t, -1)[:, s].copy_(state)
                if not self.mid_batch_reset:
                    self._prev_frozen.copy_(self._st.frozen)
            elif self._single_write:
                prob, value = self.policy.prob_value(buf.observations, self._step_rows[s])
            else:
                prob, value = self.policy.prob_value(self.step_obs)
            self._env_step(self._state, ro, prob, value, self._uniforms[s], s, self.mid_batch_reset,
                           single_write=self._single_write)
            if self._recurrent:
                # step_buf.reset -> policy.reset_one before the next serve (worker.py:46,88; sampler.py:135-138)
                hit = self._st.reset_flag if self.mid_batch_reset else \
                    self._st.frozen * (1 - self._prev_frozen)
                self.policy.reset_rows(hit)
        if self.need_extra_obs and not self._extra_is_step_obs:
            _lib.copy_bytes(buf.extra_observations, self.step_obs)     # sampler.py:147-151 (a kernel node, 16-byte copies)
        if not self.mid_batch_reset:                           # worker.py:108-113
            _lib.env_reset(self._game, self._state, ro, self._st.frozen, env.max_start_noops)
        _lib.copy_bytes(cur.host_block, self._results_block)

    def _kernel_max_path_length(self):
        """The kernels end an episode when Length > limit (worker.py:42)."""
        return self.max_path_length

    def _env_step(self, state, ro, prob, value, uniforms, step, mid_batch_reset, active=None, single_write=False,
                  limit=None):
        """The env side of one agent step: ONE launch (arl_env_step); the two-launch form only for a limit below
        one step, which the fused kernel's reset forecast does not cover.  limit: episodes end when Length > limit
        (default: this sampler family's rule for served steps)."""
        if limit is None:
            limit = self._kernel_max_path_length()
        if limit >= 1:
            _lib.env_step(self._game, state, ro, prob, value, uniforms, step, mid_batch_reset, limit,
                          self.discount, self.env.max_start_noops, active=active, single_write=single_write)
        else:
            _lib.env_act_step(self._game, state, ro, prob, value, uniforms, step, mid_batch_reset, limit,
                              self.discount, active=active)
            _lib.env_frame_step(self._game, state, ro, step, self.env.max_start_noops)

    def _capture(self, cur):
        """Warm up on a side stream, then capture one batch into a hipGraph (one graph per host-side buffer set)."""
        snap = {k: v.clone() for k, v in self._st.items()}
        obs_snap = self.step_obs.clone()
        hidden_snap = [h.clone() for h in self.policy.get_prev_hiddens()] if self._recurrent else []
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._enqueue_batch(cur)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode=graph_capture_mode()):
            self._enqueue_batch(cur)
        # warm-up and capture-time work must not count: restore the state
        for k, v in snap.items():
            self._st[k].copy_(v)
        self.step_obs.copy_(obs_snap)
        for h, v in zip(self.policy.get_prev_hiddens() if self._recurrent else [], hidden_snap):
            h.copy_(v)
        torch.cuda.synchronize(self.device)
        cur.graph = graph

    def _drain_traj_infos(self, cur):
        """Completed episodes of the batch whose event has passed (the reference's traj_infos_queue),
        read from the pinned mirrors of its buffer set; tops the no-op ring up from the same snapshot."""
        cur.event.synchronize()
        h = cur.host
        if int(h.epoch[2]):
            raise RuntimeError("arl_env_step: %d mid-batch resets were not announced by the previous launch's forecast "
                               "(st.next_reset): the start no-op draws of their RNG streams are misordered.  The length "
                               "limit changed between launches, or the emulator's termination depends on the action"
                               % int(h.epoch[2]))
        count = min(int(h.done_count[0]), self._state.done_capacity)
        infos = []
        if count:
            ints, flts = h.done_int[:count].numpy(), h.done_flt[:count].numpy()
            for (env_id, length, nonzero), (ret, raw, disc) in zip(ints, flts):
                infos.append(TrajInfo(Length=int(length), Return=float(ret), RawReturn=float(raw),
                                      NonzeroRewards=int(nonzero), DiscountedReturn=float(disc),
                                      _env=int(env_id)))
            self._refill_noop_ring(2 * self.n_parallel, int(h.epoch[0]), h.noop_cursor.numpy())
        return infos

    def _refill_noop_ring(self, n_streams, epoch=None, cursors=None):
        """Top the start-noop ring up with fresh draws from each worker stream.  Without a snapshot
        (start-up paths) the cursors are read from the device, which waits for the stream."""
        if self.env.max_start_noops <= 0:
            return
        if epoch is None:
            epoch = int(self._st.epoch[0].item())
            cursors = self._st.noop_cursor.cpu().numpy()
        cursor = cursors[epoch & 1]
        dirty = False
        for w in range(n_streams):
            n_new = int(cursor[w] + NOOP_RING - self._ring_produced[w])
            if n_new >= NOOP_RING // 2:
                draws = synth.draw_noops(self._worker_rngs[w], self.env.max_start_noops, size=n_new)
                pos = (self._ring_produced[w] + np.arange(n_new)) % NOOP_RING
                self._ring_host[w, pos] = draws
                self._ring_produced[w] += n_new
                dirty = True
        if dirty:
            self._st.noop_ring.copy_(torch.from_numpy(self._ring_host))

    def _decorrelate(self):
        """sampler/util.py:34-57: before the first batch every env takes a random number
        (< max_decorrelation_steps) of random-action steps, resetting at episode ends.
        The reference derives the count from wall-clock digits (non-deterministic); here
        counts and actions come from a RandomState derived from the sampler seed."""
        n, env, dev = self._total_n_envs, self.env, self.device
        n_act = self.env_spec.action_space.n
        rs = np.random.RandomState((self.seed + 7919) % 4294967294)
        counts = torch.from_numpy((rs.rand(n) * self.max_decorrelation_steps).astype(np.int64)).to(dev)
        gen = torch.Generator(device=dev)
        gen.manual_seed(int(rs.randint(2 ** 31)))
        scratch = struct(
            observations=self.step_obs, rewards=torch.zeros(n, device=dev),
            dones=torch.zeros(n, dtype=torch.uint8, device=dev),
            env_infos=dict(raw_reward=torch.zeros(n, device=dev),
                           need_reset=torch.zeros(n, dtype=torch.uint8, device=dev)),
            actions=torch.zeros(n, dtype=torch.uint8, device=dev),
            agent_infos=dict(prob=torch.zeros((n, n_act), device=dev), value=torch.zeros(n, device=dev)))
        ro = self._make_rollout(scratch)
        ro.horizon = 1
        prob = torch.full((n, n_act), 1. / n_act, dtype=torch.float32, device=dev)
        value = torch.zeros(n, dtype=torch.float32, device=dev)
        for k in range(int(counts.max().item()) if n else 0):
            active = (counts > k).to(torch.uint8)
            u = torch.rand(n, dtype=torch.float64, device=dev, generator=gen)
            # (the start-up walk resets at Length > max_path_length in every sampler family: sampler/util.py:50)
            self._env_step(self._state, ro, prob, value, u, 0, True, active=active, limit=self.max_path_length)
            if k % 256 == 255:
                self._st.done_count.zero_()
                self._refill_noop_ring(2 * self.n_parallel)
        self._st.done_count.zero_()
        self._refill_noop_ring(2 * self.n_parallel)
        served = self._kernel_max_path_length()
        if served != self.max_path_length:
            # The walk's last launch forecast each env's next reset under the walk's rule (Length > L); the served steps
            # of this sampler family end an episode at Length >= L (worker_with_eval.py:48): an env that leaves the walk
            # one step short of L resets in the FIRST served launch, and that launch ranks the stream's no-op draws
            # from the forecast -- restate it under the served rule (only the over-length term can differ)
            par = int(self._st.launch_count.item()) & 1
            self._st.next_reset[par] |= (self._st.traj_len + 1 > served).to(torch.uint8)
