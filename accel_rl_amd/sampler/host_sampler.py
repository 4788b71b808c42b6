"""HostEnvSampler: ANY rllab-style `EnvCls` stepped on the host's cores, served from the GPU.

The reference's sampler serves whatever `Env` it is given (accel_rl/sampler/base.py:30-51;
act_server/alternating/overlap/sampler.py:40-151, overlap/worker.py:23-153): 2 * n_parallel worker processes
step `envs_per` environments each, in two groups that alternate between simulating and waiting for actions.
`GpuVecSampler` replaces all of that for environments that live on the device (`EnvCls.batched_device_env`)
and refuses every other class.  This sampler is the other half of the boundary: a real ALE `AtariEnv`, a
MuJoCo env, BASELINE config 1's "CPU sampler" form -- stepped by pinned worker processes, with everything the
learner touches in the SAME place and layout as under `GpuVecSampler`:

  * observations travel host -> device once per (step, group) out of a shared, page-locked step buffer
    (the reference's `step_buf.obs`), and are scattered on the device into the env-major rollout buffer;
  * actions are sampled ON the device from the policy's probabilities (`arl_sample_categorical`: the
    reference's `weighted_sample_n`, rllab/misc/special.py:22-27, with the master's `np.random.rand(B)` per
    (step, group) as its uniforms -- the same variates in the same order) and travel back as B bytes;
  * `samples_buf` is the same struct of device tensors (keys, dtypes, `flat index = env * horizon + t`), so
    `process_samples`, the optimizers and the runners do not know which sampler filled it.

Rewards / dones / env_infos are written by the workers into shared host arrays and uploaded once per batch
(a few bytes per step).  Worker i is seeded `seed + i` and pinned to `affinities["sim_cpus"][i]`
(overlap/sampler.py:154-185, sampler/util.py:60-72); env index e = (group * n_parallel + rank) * envs_per + i.

Round 6: what the reference's sampler serves on ANY environment is served here too --
  * recurrent policies: one hidden-state row per env on the device, rows of a group reset from the workers' `reset`
    flags before the group is served (overlap/sampler.py:135-138, policies/base.py:50-93), the previous state of every
    (env, step) stored under the policy's `state_info_keys`;
  * epsilon-greedy (DQN-family) policies: the master's draws of a whole batch -- per (step, group) `np.random.rand(B)`
    then `action_space.sample_n(#random)` (policies/dqn/atari_dqn_policy.py:123-128) -- are made up front in that order
    (`policy.host_draws`) and shipped as an override table; a group is served from its slice (`policy.serve_group`);
  * the evaluation variant (`HostEnvEvalSampler`, sampler_with_eval.py:6-54 / worker_with_eval.py:20-239): separate
    evaluation envs in every worker, `evaluate_policy(itr)`, and that family's `Length >= max_path_length` rule."""
import multiprocessing as mp
import gc
import os
import queue as pyqueue
import time

import numpy as np
import torch

from accel_rl_amd import _lib
from accel_rl_amd.buffers import (batch_buffer, buffer_length, buffer_with_segs_view, combine_distinct_buffers,
                                  count_buffer_size)
from accel_rl_amd.sampler.base import BaseMbSampler
from accel_rl_amd.sampler.util import TrajInfo
from accel_rl_amd.util import logger
from accel_rl_amd.util.misc import nbytes_unit, struct
from accel_rl_amd.util.seed import set_seed


def _check_flat(info):
    """act_server/buffers.py:52-59: infos must be one level deep and numeric."""
    for k, v in info.items():
        if np.asarray(v).dtype == object:
            raise TypeError("Unsupported infos data type under key: {}\nSampler does not permit nested dictionaries, "
                            "values must be able to cast under np.asarray() and not result in dtype=='object')".format(k))


def _shared(example, length):
    """Zeroed array of shape (length,) + shape(example) in shared memory (visible to forked workers)."""
    v = np.asarray(example)
    if v.dtype == object:
        raise TypeError("Unsupported buffer example data type {}".format(v.dtype))
    t = torch.zeros((length,) + v.shape, dtype=torch.from_numpy(np.zeros(1, v.dtype)).dtype).share_memory_()
    return t


class _Running(struct):
    """Per-episode accumulators of one environment (sampler/util.py:75-101): plain Python arithmetic on whatever the
    environment returns, exactly as the reference does it (the result types follow numpy's promotion rules there too)."""

    def __init__(self, discount):
        super().__init__(Length=0, Return=0, RawReturn=0, NonzeroRewards=0, DiscountedReturn=0)
        self._discount = discount
        self._cur_discount = 1

    def step(self, r, env_info):
        self.Length += 1
        self.Return += r
        self.RawReturn += env_info.get("raw_reward", r)
        self.NonzeroRewards += r != 0
        self.DiscountedReturn += self._cur_discount * r
        self._cur_discount *= self._discount

    def finished(self):
        return dict(Length=int(self.Length), Return=float(self.Return), RawReturn=float(self.RawReturn),
                    NonzeroRewards=int(self.NonzeroRewards), DiscountedReturn=float(self.DiscountedReturn))


class _Quit(Exception):
    pass


def _worker(w, cfg, ctrl, gate, step, eval_step, batch, done_queue):
    """One simulation process: `envs_per` environments of group w.group, slots [lo, lo + envs_per) of the group's step
    buffer, rows [(env) * horizon ...) of the batch arrays.  reference: sampling_process + the two collectors,
    overlap/worker.py:23-153 (worker_with_eval.py:20-239 with evaluation envs); start_envs / initialize_worker,
    sampler/util.py:26-72."""
    code = 0

    def wait(sem):                      # every hand-off re-checks the quit flag: shutdown releases all of them
        sem.acquire()
        if ctrl.quit.value:
            raise _Quit()
    try:
        if w.cpu is not None:
            try:
                os.sched_setaffinity(0, [int(w.cpu)])
            except OSError:
                pass
        set_seed(w.seed, device_generators=False)           # (a forked child: see set_seed)
        T, per, horizon_limit, discount = cfg.horizon, cfg.envs_per, cfg.max_path_length, cfg.discount
        envs = [cfg.EnvCls(**cfg.env_args) for _ in range(per)]
        eval_envs = [cfg.EnvCls(**cfg.env_args) for _ in range(cfg.eval_envs_per)]       # worker_with_eval.py:200
        # served steps end an episode at Length > limit (worker.py:42), the evaluation family at >= (worker_with_eval.py:48)
        over_length = (lambda run: run.Length >= horizon_limit) if cfg.length_ge else (lambda run: run.Length > horizon_limit)
        obs_np, act_np, reset_np, live_np = (step.obs.numpy(), step.act.numpy(), step.reset.numpy(), step.live.numpy())
        rew_np, done_np = batch.rewards.numpy(), batch.dones.numpy()
        info_np = {k: v.numpy() for k, v in batch.env_infos.items()}
        count_np = batch.completed.numpy()
        lo, row0 = w.rank * per, w.first_env * T
        if eval_envs:
            eobs_np, eact_np, elo = eval_step.obs.numpy(), eval_step.act.numpy(), w.rank * cfg.eval_envs_per

        # -- start_envs: reset everything, optionally walk a random number of random steps
        running = [_Running(discount) for _ in envs]
        walk = np.random.RandomState((w.seed + 7919) % 4294967294)    # (the reference takes wall-clock digits here)
        for i, env in enumerate(envs):
            o = env.reset()
            n_steps = int(walk.rand() * cfg.max_decorrelation_steps) if cfg.max_decorrelation_steps else 0
            if n_steps:
                space = env.action_space
                actions = space.sample_n(n_steps) if hasattr(space, "sample_n") else [space.sample() for _ in range(n_steps)]
                for a in actions:
                    o, r, d, info = env.step(a)
                    running[i].step(r, info)
                    if running[i].Length > horizon_limit or (d and info.get("need_reset", True)):
                        o = env.reset()
                        running[i] = _Running(discount)
            obs_np[lo + i] = o
        gate.obs_ready.release()                         # start_envs done (the reference's first barrier_out)

        def collect_eval():
            """collect_eval (worker_with_eval.py:86-113, identical in both collectors): nothing is stored."""
            count_np[w.index] = 0
            for i, env in enumerate(eval_envs):
                eobs_np[elo + i] = env.reset()
            gate.obs_ready.release()
            runs = [_Running(discount) for _ in eval_envs]
            n_completed = 0
            for s in range(cfg.eval_horizon):
                wait(gate.act_ready)
                for i, env in enumerate(eval_envs):
                    o, r, d, info = env.step(eact_np[elo + i])
                    runs[i].step(r, info)
                    if runs[i].Length >= horizon_limit or (d and info.get("need_reset", True)):
                        o = env.reset()
                        done_queue.put(runs[i].finished())
                        n_completed += 1
                        runs[i] = _Running(discount)
                    eobs_np[elo + i] = o
                if s == cfg.eval_horizon - 1:
                    count_np[w.index] = n_completed
                gate.obs_ready.release()
            wait(gate.ack)

        frozen = [False] * per
        while True:
            wait(gate.go)                                # a batch (or an evaluation) begins, or the sampler shuts down
            if ctrl.do_eval.value:
                collect_eval()
                continue
            gate.obs_ready.release()                     # the step buffer holds this batch's first observations
            n_completed = 0
            frozen = [False] * per
            for s in range(T):
                wait(gate.act_ready)
                for i, env in enumerate(envs):
                    if frozen[i]:
                        live_np[lo + i] = 0
                        continue
                    row = row0 + i * T + s
                    o, r, d, info = env.step(act_np[lo + i])
                    run = running[i]
                    run.step(r, info)
                    over = over_length(run)
                    live = True
                    if over or (d and info.get("need_reset", True)):
                        d = True
                        reset_np[lo + i] = True          # (policy.reset_one for recurrence)
                        if over and "need_reset" in info:
                            info["need_reset"] = True
                        done_queue.put(run.finished())
                        n_completed += 1
                        running[i] = _Running(discount)
                        if cfg.mid_batch_reset:
                            o = env.reset()
                        else:
                            frozen[i], live = True, False    # keeps its last observation until the batch ends
                    if live:
                        obs_np[lo + i] = o
                    live_np[lo + i] = 1 if live else 0
                    rew_np[row] = r
                    done_np[row] = d
                    for k, v in info.items():
                        info_np[k][row] = v
                if s == T - 1:
                    count_np[w.index] = n_completed      # (visible before the last observation is announced)
                gate.obs_ready.release()
            wait(gate.ack)                               # the master has taken the last observations and the batch arrays
            if not cfg.mid_batch_reset:                  # NonResetCollector.reset_needed_envs
                for i, env in enumerate(envs):
                    if frozen[i]:
                        obs_np[lo + i] = env.reset()
    except _Quit:
        pass
    except BaseException:      # noqa: BLE001
        import traceback
        traceback.print_exc()
        code = 1
    finally:
        import sys
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(code)


class HostEnvSampler(BaseMbSampler):

    _length_ge = False          # the served steps' over-length rule: Length > limit (worker.py:42); the eval family: >=
    eval_envs_per = 0           # (HostEnvEvalSampler: evaluation envs per worker)
    eval_horizon = 0

    def __init__(self, n_parallel=1, envs_per=1, device=None, **kwargs):
        super().__init__(n_parallel=n_parallel, envs_per=envs_per, **kwargs)
        self._total_n_envs = 2 * n_parallel * envs_per
        self.device = device
        self.workers = []
        self._registered = []

    # ------------------------------------------------------------------ API
    def initialize(self, seed, affinities=None, discount=1, need_extra_obs=False):
        """reference: overlap/sampler.py:40-79"""
        if getattr(self.EnvCls, "batched_device_env", False):
            raise TypeError("HostEnvSampler steps host environments; {} lives on the device: use GpuVecSampler".format(self.EnvCls))
        affinities = affinities or dict()
        n, t, half = self._total_n_envs, self.horizon, self.n_parallel * self.envs_per
        self.seed, self.need_extra_obs = seed, need_extra_obs
        self.discount = 1. if discount is None else discount
        self.sample_size = n * t

        # -- example env, example transition (act_server/buffers.py:7-21): the same global-RNG draws as the reference
        env = self.EnvCls(**self.env_args)
        env.reset()
        obs, reward, done, env_info = env.step(env.spec.action_space.sample())
        _check_flat(env_info)
        self.env, self.env_spec = env, env.spec
        self._examples = dict(observations=np.asarray(obs), rewards=reward, dones=done, env_infos=dict(env_info))

        # -- shared host side: per-batch scalars + one step buffer per group (build_step_buffer, :24-30), then the
        #    evaluation step buffers (build_par_objs, sampler.py:204-206) -- the example draws in the reference's order
        self._batch = struct(rewards=_shared(reward, n * t), dones=_shared(done, n * t),
                             env_infos={k: _shared(v, n * t) for k, v in env_info.items()},
                             completed=_shared(np.int64(0), 2 * self.n_parallel))     # episodes each worker finished
        self._steps, self._eval_steps = [], []
        for _ in range(2):
            ex_obs, ex_act = env.spec.observation_space.sample(), env.spec.action_space.sample()
            self._steps.append(struct(obs=_shared(ex_obs, half), act=_shared(ex_act, half), reset=_shared(False, half),
                                      live=_shared(np.uint8(0), half)))
        half_e = self.n_parallel * self.eval_envs_per
        for _ in range(2 if self.eval_envs_per else 0):
            ex_obs, ex_act = env.spec.observation_space.sample(), env.spec.action_space.sample()
            self._eval_steps.append(struct(obs=_shared(ex_obs, half_e), act=_shared(ex_act, half_e)))
        self._act_dtype, self._act_np_dtype = self._steps[0].act.dtype, np.asarray(ex_act).dtype

        # -- the processes (forked BEFORE this process touches the GPU for this sampler; they never do)
        ctx = mp.get_context("fork")
        # hand-offs are semaphores only (the reference pairs them with two barriers): every wait of the master can then
        # poll its workers' health instead of blocking for ever on one that died
        self._ctrl = struct(quit=ctx.RawValue("b", 0), do_eval=ctx.RawValue("b", 0))
        self._gates = [[struct(go=ctx.Semaphore(0), obs_ready=ctx.Semaphore(0), act_ready=ctx.Semaphore(0),
                               ack=ctx.Semaphore(0)) for _ in range(self.n_parallel)] for _ in range(2)]
        self._done_queue = ctx.Queue()
        cfg = struct(EnvCls=self.EnvCls, env_args=self.env_args, envs_per=self.envs_per, horizon=t,
                     max_path_length=self.max_path_length, discount=self.discount, mid_batch_reset=self.mid_batch_reset,
                     max_decorrelation_steps=self.max_decorrelation_steps, length_ge=self._length_ge,
                     eval_envs_per=self.eval_envs_per, eval_horizon=self.eval_horizon)
        cpus = affinities.get("sim_cpus") if hasattr(affinities, "get") else None
        i = 0
        # The workers are FORKED from a process whose HIP runtime may be up.  A child must never finalise an object of the
        # parent's that owns device state (a captured graph, an event, page-locked memory left in an unreachable cycle):
        # its destructor would call into a runtime the child does not have -- a segfault inside the child's first garbage
        # collection, one worker start in a few hundred (tests/test_host_sampler_gpu.py after 200 other GPU tests).  So the
        # parent collects its garbage HERE, and everything alive at the fork is frozen out of the children's collector.
        gc.collect()
        gc.freeze()
        try:
            for group in range(2):
                for rank in range(self.n_parallel):
                    w = struct(group=group, rank=rank, index=i, seed=seed + i,
                               first_env=(group * self.n_parallel + rank) * self.envs_per,
                               cpu=cpus[i] if cpus is not None and i < len(cpus) else None)
                    p = ctx.Process(target=_worker, args=(w, cfg, self._ctrl, self._gates[group][rank], self._steps[group],
                                                          self._eval_steps[group] if self._eval_steps else None,
                                                          self._batch, self._done_queue), daemon=True)
                    p.start()
                    self.workers.append(p)
                    i += 1
        finally:
            gc.unfreeze()

        # -- device side: the SAME buffers GpuVecSampler fills
        _lib.load()
        if self.device is None:
            self.device = torch.device("cuda", int(affinities.get("gpu", 0) or 0))
        dev = self.device = torch.device(self.device)
        self.envs_buf = buffer_with_segs_view(self._examples, n * t, t, dev)
        if need_extra_obs:
            self.envs_buf.extra_observations = batch_buffer(self._examples["observations"], n, dev)
        self.step_obs = batch_buffer(self._examples["observations"], n, dev)
        if self.eval_envs_per:
            self.eval_step_obs = batch_buffer(self._examples["observations"], 2 * half_e, dev)
        self._pin_shared()
        return self.env_spec, self.sample_size, self.horizon, self.mid_batch_reset

    def policy_init(self, policy):
        """reference: overlap/sampler.py:81-95"""
        dev, n, t, half = self.device, self._total_n_envs, self.horizon, self.n_parallel * self.envs_per
        self.policy = policy
        self._recurrent = bool(getattr(policy, "recurrent", False))
        self._eps_greedy = hasattr(policy, "host_draws")
        if self._recurrent and self.eval_envs_per:
            # the reference serves its evaluation groups through the SAME hidden-state pair as the training groups
            # (sampler_with_eval.py:36-50 calls policy.get_actions on eval_step_bufs): with eval_envs_per != envs_per
            # the state's row count does not match and Theano raises; with equal counts evaluation silently overwrites
            # the training envs' hidden states.  Neither is a behaviour to reproduce.
            raise NotImplementedError("recurrent policies with the evaluation sampler (the reference's own combination "
                                      "does not run: INTEGRATION.md, section E)")
        if self._recurrent and not hasattr(policy, "act_step"):
            raise NotImplementedError("recurrent policies must provide act_step(observations, rows=(lo, hi)) / reset_rows")
        if self._eps_greedy and not hasattr(policy, "serve_group"):
            raise NotImplementedError("policies with their own action draws must provide serve_group (QPolicyBase)")
        n_act = self.env_spec.action_space.n
        policy.reset(n_batch=1)                           # build_policy_buffer (act_server/buffers.py:33-38)
        policy.get_action(torch.from_numpy(np.asarray(self.env_spec.observation_space.sample())).to(dev))
        agent_infos = dict(prob=np.zeros(n_act, np.float32), value=np.float32(0))
        if self._recurrent:                               # previous hidden state of every (env, step) (policies/base.py:86-93)
            for key, state in zip(policy.state_info_keys, policy.get_prev_hiddens()):
                agent_infos[key] = np.zeros(state.shape[1], np.float32)
        examples = dict(actions=np.zeros((), self._act_np_dtype), agent_infos=agent_infos)
        policy_buf = buffer_with_segs_view(examples, n * t, t, dev)
        self.samples_buf = combine_distinct_buffers(self.envs_buf, policy_buf)
        assert buffer_length(self.samples_buf) == self.sample_size
        # recurrent: ONE state row per env, group j = rows [j * half, (j + 1) * half) -- the reference's pair of
        # per-group states (policies/base.py:44-93) side by side, which is also what its get_state_info() returns
        policy.reset(n_batch=n if self._recurrent else half)
        widest = max(half, self.n_parallel * self.eval_envs_per)
        self._uniforms_host = torch.empty(widest, dtype=torch.float64).pin_memory()
        self._uniforms = torch.empty(widest, dtype=torch.float64, device=dev)
        self._act_dev = torch.empty(widest, dtype=torch.uint8, device=dev)
        self._live_dev = torch.empty(half, dtype=torch.uint8, device=dev)
        if self._recurrent:
            self._reset_host = torch.zeros(n, dtype=torch.uint8).pin_memory()
            self._reset_dev = torch.zeros(n, dtype=torch.uint8, device=dev)
        for j in range(2):
            self._acquire(j, "obs_ready")                 # start_envs done everywhere
        logger.log("HostEnvSampler -- total_n_envs: {}".format(self.total_n_envs))
        logger.log("HostEnvSampler -- batch buffer size: {:,.1f} {}".format(*nbytes_unit(count_buffer_size(self.samples_buf))))

    def _serve(self, obs_dev, s, lo, n_total, width):
        """One serving call of the master (policy.get_actions, overlap/sampler.py:139): policy outputs for the `width`
        observations of one group, then the group's actions sampled on the device into self._act_dev[:width].
        Returns (prob, value, previous recurrent state or ())."""
        policy, prev = self.policy, ()
        if self._eps_greedy:                              # the batch's draws were made up front (host_draws)
            policy.set_step(s)
            prob, value = policy.serve_group(obs_dev, lo, n_total)
            self._uniforms_host[:width].fill_(0.5)        # a one-hot row: any uniform selects the hot action
        else:
            if self._recurrent:
                prob, value, *prev = policy.act_step(obs_dev, rows=(lo, lo + width))
            else:
                prob, value = policy.prob_value(obs_dev)
            self._uniforms_host[:width].copy_(torch.from_numpy(np.random.rand(width)))     # special.py:24
        self._uniforms[:width].copy_(self._uniforms_host[:width], non_blocking=True)
        _lib.sample_categorical(prob, self._uniforms[:width], self._act_dev[:width])
        return prob, value, prev

    def obtain_samples(self, itr):
        """reference: overlap/sampler.py:97-104 + serve_actions :120-151"""
        n, t, half = self._total_n_envs, self.horizon, self.n_parallel * self.envs_per
        buf, dev = self.samples_buf, self.device
        obs_rows = buf.observations.view((n, t) + tuple(buf.observations.shape[1:]))
        act_rows, prob_rows, value_rows = (buf.actions.view(n, t), buf.agent_infos["prob"].view(n, t, -1),
                                           buf.agent_infos["value"].view(n, t))
        if self._eps_greedy:                              # per (step, group): rand(B), then sample_n(#random) -- all of the
            self.policy.host_draws(t, n)                  # batch's, in the reference's order (atari_dqn_policy.py:123-128)
        for gates in self._gates:
            for g in gates:
                g.go.release()
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream()
            for s in range(t):
                for j in range(2):
                    lo = j * half
                    self._acquire(j, "obs_ready")
                    step = self._steps[j]
                    obs_dev = self.step_obs[lo:lo + half]
                    obs_dev.copy_(step.obs, non_blocking=True)
                    if self._recurrent and bool(step.reset.any()):     # policy.reset_one for the group's flagged envs
                        self._reset_host.zero_()                       # (sampler.py:135-138)
                        self._reset_host[lo:lo + half].copy_(step.reset.to(torch.uint8))
                        self._reset_dev.copy_(self._reset_host, non_blocking=True)
                        self.policy.reset_rows(self._reset_dev)
                    prob, value, prev = self._serve(obs_dev, s, lo, n, half)
                    act_dev = self._act_dev[:half]
                    if self.mid_batch_reset or s == 0:
                        live = None
                    else:
                        self._live_dev.copy_(step.live, non_blocking=True)
                        live = self._live_dev
                    step.act.copy_(act_dev.to(self._act_dtype) if self._act_dtype != torch.uint8 else act_dev,
                                   non_blocking=True)
                    stream.synchronize()                  # the actions are in the step buffer
                    step.reset.zero_()
                    for g in self._gates[j]:
                        g.act_ready.release()
                    # scatter (on the device, under the workers' simulation): the observation served at step s is row
                    # (env, s); an env that froze (mid_batch_reset=False) keeps the rows of the batch before
                    if live is None:
                        obs_rows[lo:lo + half, s] = obs_dev
                    else:
                        keep = torch.nonzero(live).squeeze(1)
                        obs_rows[lo + keep, s] = obs_dev[keep]
                    act_rows[lo:lo + half, s] = act_dev.to(act_rows.dtype)
                    prob_rows[lo:lo + half, s] = prob
                    value_rows[lo:lo + half, s] = value
                    for key, state in zip(self.policy.state_info_keys if self._recurrent else (), prev):
                        buf.agent_infos[key].view(n, t, -1)[lo:lo + half, s] = state
            for j in range(2):
                self._acquire(j, "obs_ready")
                if self.need_extra_obs:
                    buf.extra_observations[j * half:(j + 1) * half].copy_(self._steps[j].obs, non_blocking=True)
            buf.rewards.copy_(self._batch.rewards, non_blocking=True)
            buf.dones.copy_(self._batch.dones, non_blocking=True)
            for k, v in self._batch.env_infos.items():
                buf.env_infos[k].copy_(v, non_blocking=True)
            stream.synchronize()                          # everything the workers own has been read ...
        return self.samples_buf, self._finish_round()

    def _finish_round(self):
        """The workers may go on (reset frozen envs, wait for the next round); their completed trajectories."""
        n_done = int(self._batch.completed.sum())
        for gates in self._gates:
            for g in gates:
                g.ack.release()
        infos = []
        while len(infos) < n_done:                        # the reference's traj_infos_queue, without its qsize() race
            try:
                infos.append(TrajInfo(**self._done_queue.get(timeout=1.0)))
            except pyqueue.Empty:
                self._alive()
        return infos

    def shutdown(self):
        """Every hand-off a worker can be blocked in is released with the quit flag up; ONE shared deadline for the
        joins (a worker that sits in an env's own code is terminated, not waited for one after another)."""
        if not self.workers:
            return
        self._ctrl.quit.value = 1
        for gates in self._gates:
            for g in gates:
                for sem in (g.go, g.act_ready, g.ack):
                    sem.release()
        deadline = time.time() + 5.0
        for p in self.workers:
            p.join(max(0.0, deadline - time.time()))
        self.kill_workers()
        rt = torch.cuda.cudart()
        for ptr in self._registered:
            rt.cudaHostUnregister(ptr)
        self._registered = []

    def kill_workers(self):
        """Failure paths (a runner leaving through os._exit): end the simulation processes now."""
        for p in self.workers:
            if p.is_alive():
                p.terminate()
        deadline = time.time() + 2.0
        for p in self.workers:
            p.join(max(0.0, deadline - time.time()))
        self.workers = []

    @property
    def alternating(self):
        return True

    # -------------------------------------------------------------- internals
    def _pin_shared(self):
        """Page-lock the shared step / batch arrays (hipHostRegister): host <-> device copies then run as plain DMA
        without a staging copy.  A refusal is not an error: the copies still work, through pageable memory."""
        rt = torch.cuda.cudart()
        tensors = [self._batch.rewards, self._batch.dones] + list(self._batch.env_infos.values())
        for st in self._steps:
            tensors += [st.obs, st.act, st.live]
        for st in self._eval_steps:
            tensors += [st.obs, st.act]
        for tns in tensors:
            if tns.numel() == 0:
                continue
            err = rt.cudaHostRegister(tns.data_ptr(), tns.numel() * tns.element_size(), 0)
            if int(err) == 0:
                self._registered.append(tns.data_ptr())
            else:
                logger.log("HostEnvSampler: page-locking a shared buffer failed (%s); copies go through pageable memory" % err)
                break

    def _alive(self):
        dead = [i for i, p in enumerate(self.workers) if p.exitcode is not None]
        if dead:
            raise RuntimeError("HostEnvSampler: simulation worker(s) %s ended (see their traceback above)" % dead)

    def _acquire(self, group, which):
        for g in self._gates[group]:
            while not g[which].acquire(timeout=1.0):
                self._alive()


class HostEnvEvalSampler(HostEnvSampler):
    """The reference's AAOEvalSampler (sampler_with_eval.py:6-54) on host environments: every worker also holds
    `eval_envs_per` evaluation envs (constructed after its training envs, sharing its RNG stream:
    worker_with_eval.py:199-200); `evaluate_policy(itr)` resets them, serves `eval_steps // n_eval_envs` steps of the
    current policy on them -- the master's `np.random.rand(B)` per (step, group) as ever -- stores nothing and returns the
    completed TrajInfos.  Episodes of this family end at Length >= max_path_length, training steps included
    (worker_with_eval.py:48,123,159).  The reference copies the training step buffers' observations aside around an
    evaluation (sampler_with_eval.py:22,30-31); here evaluation has step buffers of its own on host and device, so the
    training ones are never touched."""

    _length_ge = True

    def __init__(self, eval_steps, eval_envs_per, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.eval_envs_per = eval_envs_per
        self._total_n_eval_envs = eval_envs_per * self.n_parallel * 2
        self.eval_horizon = eval_steps // self._total_n_eval_envs

    def evaluate_policy(self, itr):
        """sampler_with_eval.py:20-54 + collect_eval (worker_with_eval.py:86-113)."""
        ne, te, half_e = self._total_n_eval_envs, self.eval_horizon, self.n_parallel * self.eval_envs_per
        dev = self.device
        if self._eps_greedy:
            self.policy.host_draws(te, ne)
        self._ctrl.do_eval.value = 1
        for gates in self._gates:
            for g in gates:
                g.go.release()
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream()
            for s in range(te):
                for j in range(2):
                    lo = j * half_e
                    self._acquire(j, "obs_ready")
                    step = self._eval_steps[j]
                    obs_dev = self.eval_step_obs[lo:lo + half_e]
                    obs_dev.copy_(step.obs, non_blocking=True)
                    self._serve(obs_dev, s, lo, ne, half_e)
                    act_dev = self._act_dev[:half_e]
                    step.act.copy_(act_dev.to(self._act_dtype) if self._act_dtype != torch.uint8 else act_dev,
                                   non_blocking=True)
                    stream.synchronize()
                    for g in self._gates[j]:
                        g.act_ready.release()
            for j in range(2):
                self._acquire(j, "obs_ready")
        infos = self._finish_round()
        self._ctrl.do_eval.value = 0
        return infos
