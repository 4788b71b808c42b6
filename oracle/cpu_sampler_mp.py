"""TEST / BASELINE INFRASTRUCTURE ONLY -- never imported by the product (accel_rl_amd/).

The reference's CPU sampler as it actually runs: one master that serves actions and
2 * n_parallel forked worker processes that step the environments, in two alternating
groups so that one group simulates while the other is being served
(accel_rl/sampler/act_server/alternating/overlap/sampler.py:97-151, worker.py:23-153,
sampler/util.py:26-72).  Hand-offs are semaphores over shared-memory buffers, exactly one
obs-ready / act-ready pair per worker as in the reference (sampler.py:187-225).

The per-environment arithmetic is oracle/ref_port.py (PortedAtariEnv / PortedTrajInfo,
pinned to the reference by tests/golden G6/G7); this file only adds the process
structure, so that bench.py can time "the reference's own CPU sampler on the GPU box's
host cores" (north_star) with a stated core count.  tests/test_cpu_sampler_mp.py checks
that it reproduces the sequential restatement (CpuSamplerPort) bit for bit.

Worker w = group * n_parallel + rank owns envs [w*envs_per, (w+1)*envs_per) and the RNG
stream RandomState(seed + w) (sampler/util.py:68-69).
"""
import ctypes
import multiprocessing as mp
import os

import numpy as np

from oracle import ref_port as P

OBS_SHAPE = (P.OBS_H, P.OBS_W)


def _np(raw, dtype, shape):
    return np.frombuffer(raw, dtype=dtype).reshape(shape)


def _layout(n, t, f, a):
    """name -> (ctype, numpy dtype, shape) of every shared array (buffers/batch.py:15-100,
    sampler/act_server/buffers.py:7-38)."""
    obs = (f,) + OBS_SHAPE
    return dict(
        observations=(ctypes.c_uint8, np.uint8, (n * t,) + obs),
        rewards=(ctypes.c_float, np.float32, (n * t,)),
        dones=(ctypes.c_uint8, np.uint8, (n * t,)),
        raw_reward=(ctypes.c_float, np.float32, (n * t,)),
        need_reset=(ctypes.c_uint8, np.uint8, (n * t,)),
        step_obs=(ctypes.c_uint8, np.uint8, (n,) + obs),          # step_bufs[g].obs, groups back to back
        step_act=(ctypes.c_uint8, np.uint8, (n,)),                # step_bufs[g].act
        n_completed=(ctypes.c_int64, np.int64, (n,)),             # per env: trajectories finished so far
    )


def _worker(w, cfg, raws, sem_obs, sem_act, bar_in, bar_out, quit_flag, affinity):
    """sampling_process + Reset/NonResetCollector.collect (overlap/worker.py:23-153)."""
    if affinity is not None:
        try:
            os.sched_setaffinity(0, {affinity})                   # sampler/util.py:60-67
        except OSError:
            pass
    n, t, f, a = cfg["n_envs"], cfg["horizon"], cfg["n_stack"], cfg["n_actions"]
    lay = _layout(n, t, f, a)
    sh = {k: _np(raws[k], lay[k][1], lay[k][2]) for k in lay}
    per, mid, max_len = cfg["envs_per"], cfg["mid_batch_reset"], cfg["max_path_length"]
    rng = np.random.RandomState((cfg["seed"] + w) % 4294967294)
    envs = [P.PortedAtariEnv(game=cfg["game"], rng=rng, **cfg["env_kwargs"]) for _ in range(per)]
    trajs = []
    lo = w * per
    for i, env in enumerate(envs):                                 # start_envs (no decorrelation steps)
        sh["step_obs"][lo + i] = env.reset()
        trajs.append(P.PortedTrajInfo(cfg["discount"]))
    bar_out.wait()                                                 # worker.py:143
    while True:
        bar_in.wait()
        if quit_flag.value:
            return
        frozen = [False] * per
        sem_obs.release()                                          # previous step_buf obs already written
        for i in range(per):                                       # worker.py:30-32
            sh["observations"][(lo + i) * t] = sh["step_obs"][lo + i]
        for s in range(t):
            sem_act.acquire()
            for i, env in enumerate(envs):
                e = lo + i
                if not mid and frozen[i]:
                    continue
                o, r, d, info = env.step(sh["step_act"][e])
                trajs[i].step(r, info)
                over_len = trajs[i]["Length"] > max_len
                hit = over_len or (d and info.get("need_reset", True))
                if hit:
                    d = True
                    if over_len and "need_reset" in info:
                        info["need_reset"] = True
                    sh["n_completed"][e] += 1
                    trajs[i] = P.PortedTrajInfo(cfg["discount"])
                    if mid:
                        o = env.reset()
                    else:
                        frozen[i] = True
                if mid or not hit:
                    sh["step_obs"][e] = o
                    if s < t - 1:
                        sh["observations"][e * t + s + 1] = o
                sh["rewards"][e * t + s] = r
                sh["dones"][e * t + s] = d
                if "raw_reward" in info:
                    sh["raw_reward"][e * t + s] = info["raw_reward"]
                if "need_reset" in info:
                    sh["need_reset"][e * t + s] = info["need_reset"]
            sem_obs.release()
        bar_out.wait()
        if not mid:                                                # reset_needed_envs, worker.py:108-113
            for i, env in enumerate(envs):
                if frozen[i]:
                    sh["step_obs"][lo + i] = env.reset()


class CpuSamplerMP(object):
    """Master side (overlap/sampler.py:40-151)."""

    def __init__(self, game, horizon, n_parallel=1, envs_per=1, max_path_length=np.inf,
                 mid_batch_reset=True, env_kwargs=None, start_method="spawn", pin=False):
        self.game, self.horizon = game, horizon
        self.n_parallel, self.envs_per = n_parallel, envs_per
        self.max_path_length, self.mid_batch_reset = max_path_length, mid_batch_reset
        self.env_kwargs = dict(env_kwargs or {})
        self.n_envs = 2 * n_parallel * envs_per
        self.half = n_parallel * envs_per
        self._ctx = mp.get_context(start_method)
        self._pin = pin
        self._procs = []

    def initialize(self, seed, discount=1., master_rng=None):
        mrng = np.random if master_rng is None else master_rng
        example = P.PortedAtariEnv(game=self.game, rng=mrng, **self.env_kwargs)     # sampler.py:42
        example.reset()
        example.step(int(mrng.randint(example.n_actions, dtype=np.uint8)))
        n, t = self.n_envs, self.horizon
        self.n_actions, f = example.n_actions, example.n_stack
        self._cfg = dict(n_envs=n, horizon=t, n_stack=f, n_actions=self.n_actions, envs_per=self.envs_per,
                         mid_batch_reset=self.mid_batch_reset, max_path_length=self.max_path_length,
                         seed=seed, game=self.game, env_kwargs=self.env_kwargs, discount=discount)
        lay = _layout(n, t, f, self.n_actions)
        self._raws = {k: self._ctx.RawArray(c, int(np.prod(shape))) for k, (c, d, shape) in lay.items()}
        self._sh = {k: _np(self._raws[k], lay[k][1], lay[k][2]) for k in lay}
        self.buf = dict(
            observations=self._sh["observations"], rewards=self._sh["rewards"],
            dones=self._sh["dones"].view(bool), raw_reward=self._sh["raw_reward"],
            need_reset=self._sh["need_reset"].view(bool),
            actions=np.zeros(n * t, np.uint8), prob=np.zeros((n * t, self.n_actions), np.float32),
            value=np.zeros(n * t, np.float32), extra_observations=np.zeros((n, f) + OBS_SHAPE, np.uint8))
        n_workers = 2 * self.n_parallel
        self._quit = self._ctx.RawValue(ctypes.c_int, 0)
        self._bar_in = self._ctx.Barrier(n_workers + 1)            # ctrl.barrier_in / barrier_out
        self._bar_out = self._ctx.Barrier(n_workers + 1)
        self._sem_obs = [self._ctx.Semaphore(0) for _ in range(n_workers)]     # sync.step_blockers
        self._sem_act = [self._ctx.Semaphore(0) for _ in range(n_workers)]     # sync.act_waiters
        cpus = sorted(os.sched_getaffinity(0))
        # (forked from a process that may hold device objects -- bench.py's cpu_baseline leg --: none of them may be
        #  finalised by a child's garbage collector; see accel_rl_amd/sampler/host_sampler.py)
        import gc
        gc.collect()
        gc.freeze()
        try:
            for w in range(n_workers):
                aff = cpus[(w + 1) % len(cpus)] if self._pin else None
                p = self._ctx.Process(target=_worker, args=(w, self._cfg, self._raws, self._sem_obs[w],
                                                            self._sem_act[w], self._bar_in, self._bar_out,
                                                            self._quit, aff), daemon=True)
                p.start()
                self._procs.append(p)
        finally:
            gc.unfreeze()
        self._bar_out.wait(timeout=120)                            # envs are started
        self._completed_seen = 0
        return self.n_actions, n * t

    def _workers_of(self, group):
        return range(group * self.n_parallel, (group + 1) * self.n_parallel)

    def obtain_samples(self, policy):
        """sampler.py:97-104,120-151.  Returns (buffers, number of trajectories completed)."""
        n, t, half, b = self.n_envs, self.horizon, self.half, self.buf
        self._bar_in.wait(timeout=120)
        for s in range(t):
            for j in (0, 1):
                for w in self._workers_of(j):
                    if not self._sem_obs[w].acquire(timeout=120):
                        raise RuntimeError("CPU sampler worker %d did not report its observations" % w)
                lo, hi = j * half, (j + 1) * half
                acts, infos = policy.get_actions(self._sh["step_obs"][lo:hi])
                self._sh["step_act"][lo:hi] = acts
                for w in self._workers_of(j):
                    self._sem_act[w].release()
                idx = np.arange(lo, hi) * t + s
                b["actions"][idx] = acts
                b["prob"][idx] = infos["prob"]
                b["value"][idx] = infos["value"]
        for w in range(2 * self.n_parallel):                       # last step done everywhere
            self._sem_obs[w].acquire()
        b["extra_observations"][:] = self._sh["step_obs"]          # sampler.py:147-151
        self._bar_out.wait()
        total = int(self._sh["n_completed"].sum())
        new = total - self._completed_seen
        self._completed_seen = total
        return b, new

    def shutdown(self):
        if not self._procs:
            return
        self._quit.value = 1
        self._bar_in.wait()
        for p in self._procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        self._procs = []
