"""Single-GPU training loop.

Reference: accel_rl/runners/accel_rl_base.py:15-150 (AccelRLBase) and
accel_rl/runners/accel_rl.py:12-104 (AccelRL).  Same constructor, seeds
(runner `seed`, sampler `seed + 1`), iteration arithmetic, logged keys and
`SamplesPerSecond` definition (accel_rl.py:93-98) -- the metric BASELINE.json
quotes.  Differences forced by the device-resident design: diagnostics that the
reference computes from host arrays every iteration (entropy / perplexity EMA)
are kept as device scalars and only read at log time, so an iteration issues no
host<->device sync besides the sampler's trajectory-info drain.
"""
import time
from collections import deque

import numpy as np
import torch

from accel_rl_amd.runners.base import Runner
from accel_rl_amd.util import logger
from accel_rl_amd.util.misc import make_seed, nbytes_unit
from accel_rl_amd.util.quick_args import save_args
from accel_rl_amd.util.seed import set_seed


class AccelRLBase(Runner):

    def __init__(self, algo, policy, sampler, n_steps, seed=None, affinities=None, use_gpu=True):
        n_steps = int(n_steps)
        save_args(vars(), underscore=False)
        if affinities is None:
            self.affinities = dict()
        if algo.optimizer.parallelism_tag != self.parallelism_tag:
            raise TypeError("Had mismatched parallelism between Runner ({}) and Optimizer: "
                            "{}".format(self.parallelism_tag, algo.optimizer.parallelism_tag))

    def startup(self, master=True):
        """reference: accel_rl_base.py:36-59"""
        if self.seed is None:
            self.seed = make_seed()
        set_seed(self.seed)
        env_spec, sample_size, horizon, mid_batch_reset = self.sampler.initialize(
            seed=self.seed + 1, affinities=self.affinities,
            discount=getattr(self.algo, "discount", None),
            need_extra_obs=self.algo.need_extra_obs)
        self.init_policy(env_spec)
        self.algo.initialize(policy=self.policy, env_spec=env_spec, sample_size=sample_size,
                             horizon=horizon, mid_batch_reset=mid_batch_reset)
        self.sampler.policy_init(self.policy)
        if master:
            n_itr = self.get_n_itr(sample_size)
            self.algo.set_n_itr(n_itr)
            if hasattr(self.algo, "set_log_interval_itrs"):          # (diagnostics held until the next log line)
                self.algo.set_log_interval_itrs(self._log_interval_itrs)
            self.init_logging()
            return n_itr

    def init_policy(self, env_spec):
        if not self.use_gpu:
            raise NotImplementedError("accel_rl_amd has no CPU learner: use_gpu must be True (INTEGRATION.md, section E)")
        self.policy.initialize(env_spec, device=self.sampler.device)
        logger.log("Policy trainable params -- number: {:,}   size: {:,.1f} {}".format(
            self.policy.n_params, *nbytes_unit(self.policy.n_params * 4)))
        self.pin_master()

    def pin_master(self):
        """accel_rl_base.py:71-72: the runner's process (here: the thread that replays the rank's graphs) is pinned to
        affinities["gpu_cpus"]; no key = untouched.  CPUs the process may not use (a cgroup narrower than the launcher's
        table) are dropped, and an empty remainder leaves the affinity as it is instead of failing the run."""
        import os
        want = self.affinities.get("gpu_cpus") if hasattr(self.affinities, "get") else None
        if not want:
            return None
        allowed = os.sched_getaffinity(0)
        cpus = sorted(set(int(c) for c in want) & allowed)
        if not cpus:
            logger.log("WARNING: gpu_cpus %s not among this process's CPUs %s: affinity unchanged" % (list(want), sorted(allowed)))
            return None
        os.sched_setaffinity(0, cpus)
        logger.log("Runner CPU affinity: %s" % cpus)
        return cpus

    def get_n_itr(self, sample_size):
        """reference: accel_rl_base.py:74-87"""
        self._sample_size = sample_size
        self._log_interval_itrs = max(self._log_steps // sample_size, 1)
        n_itr = max(self.n_steps // sample_size, 1)
        rem = n_itr % self._log_interval_itrs
        if rem <= self._log_interval_itrs / 2.:
            n_itr -= rem
        else:
            n_itr += self._log_interval_itrs - rem
        n_itr += 1
        self._n_itr = n_itr
        logger.log("Iterations to run: {}".format(n_itr))
        return n_itr

    def init_logging(self):
        self._opt_infos = {k: list() for k in self.algo.opt_info_keys}
        self._initial_param_vector = self.policy.flat_params.clone()
        self._start_time = self._last_time = time.time()

    def shutdown(self):
        logger.log("Training complete.")
        self.sampler.shutdown()

    def get_itr_snapshot(self, itr):
        """reference: accel_rl_base.py:108-113"""
        return dict(itr=itr, cum_samples=itr * self._sample_size,
                    policy_param_values=self.policy.get_param_values())

    def save_itr_snapshot(self, itr):
        logger.save_itr_params(itr, self.get_itr_snapshot(itr))

    def _log_infos(self, traj_infos=None):
        """reference: accel_rl_base.py:122-146"""
        traj_infos = self._traj_infos if traj_infos is None else traj_infos
        if traj_infos:
            for k in traj_infos[0]:
                if not k.startswith("_"):
                    logger.record_tabular_misc_stat(k, [info[k] for info in traj_infos])
        for k, vals in self._opt_infos.items():
            flat = []
            for v in vals:
                flat.extend(v.reshape(-1).tolist() if isinstance(v, torch.Tensor) else [v])
            logger.record_tabular_misc_stat(k, flat)
        self._opt_infos = {k: list() for k in self._opt_infos}
        p = self.policy.flat_params
        logger.record_tabular("ParamsNorm", torch.sqrt(torch.sum(p * p)).item())
        diff = p - self._initial_param_vector
        logger.record_tabular("NormFromInit", torch.sqrt(torch.sum(diff * diff)).item())

    @property
    def parallelism_tag(self):
        return "single"


class AccelRL(AccelRLBase):
    """Runs RL; tracks performance online using learning trajectories."""

    def __init__(self, log_interval_steps=1e5, log_traj_window=100, log_ema_steps=None, **kwargs):
        super().__init__(**kwargs)
        self._log_steps = int(log_interval_steps)
        self._log_traj_window = int(log_traj_window)
        self._log_ema_steps = int(log_interval_steps) if log_ema_steps is None else int(log_ema_steps)

    def train(self):
        """reference: accel_rl.py:28-37"""
        n_itr = self.startup()
        for itr in range(n_itr):
            with logger.prefix("itr #%d | " % itr):
                samples_data, traj_infos = self.sampler.obtain_samples(itr)
                opt_data, opt_infos = self.algo.optimize_policy(itr, samples_data)
                self.store_diagnostics(itr, samples_data, opt_data, traj_infos, opt_infos)
                if (itr + 1) % self._log_interval_itrs == 0:
                    self.log_diagnostics(itr)
        self.shutdown()

    def init_logging(self):
        self._traj_infos = deque(maxlen=self._log_traj_window)
        self._cum_completed_steps = 0
        self._cum_completed_trajs = 0
        self._new_completed_trajs = 0
        self._log_entropy = hasattr(self.policy, "distribution")
        if self._log_entropy:
            dev = self.policy.device
            self._entropy_ema = torch.ones((), device=dev)
            self._perplexity_ema = torch.ones((), device=dev)
            self._ema_a = 1 - (0.01) ** (self._log_ema_steps / self._sample_size)   # accel_rl.py:48
        logger.log("optimizing over {} iterations".format(self._log_interval_itrs))
        super().init_logging()

    def store_diagnostics(self, itr, samples_data, opt_data, traj_infos, opt_infos):
        """reference: accel_rl.py:55-74"""
        self._cum_completed_trajs += len(traj_infos)
        self._new_completed_trajs += len(traj_infos)
        for traj_info in traj_infos:
            self._cum_completed_steps += traj_info["Length"]
            self._traj_infos.append(traj_info)
        for k, v in opt_infos.items():
            self._opt_infos[k].extend(v if isinstance(v, list) else [v])
        if self._log_entropy:
            entropies = self.policy.distribution.entropy(samples_data.agent_infos)
            a = self._ema_a
            self._entropy_ema = a * entropies.mean() + (1 - a) * self._entropy_ema
            self._perplexity_ema = a * torch.exp(entropies).mean() + (1 - a) * self._perplexity_ema

    def log_diagnostics(self, itr):
        """reference: accel_rl.py:76-104"""
        self.save_itr_snapshot(itr)
        logger.record_tabular("Iteration", itr)
        logger.record_tabular("CumCompletedTrajs", self._cum_completed_trajs)
        logger.record_tabular("CumCompletedSteps", self._cum_completed_steps)
        logger.record_tabular("CumTotalSteps", (itr + 1) * self._sample_size)
        logger.record_tabular("NewCompletedTrajs", self._new_completed_trajs)
        logger.record_tabular("StepsInTrajWindow", sum(info["Length"] for info in self._traj_infos))
        if self._log_entropy:
            logger.record_tabular("Entropy", self._entropy_ema.item())
            logger.record_tabular("Perplexity", self._perplexity_ema.item())
        self._log_infos()
        if torch.device(self.policy.device).type == "cuda":
            torch.cuda.synchronize(self.policy.device)
        new_time = time.time()
        samples_per_second = (self._log_interval_itrs * self._sample_size) / (new_time - self._last_time)
        logger.record_tabular("CumTime (s)", new_time - self._start_time)
        logger.record_tabular("SamplesPerSecond", samples_per_second)
        self._last_time = new_time
        self.last_tabular = logger.dump_tabular(with_prefix=False)
        self._new_completed_trajs = 0
        if itr < self._n_itr - 1:
            logger.log("optimizing over {} iterations".format(self._log_interval_itrs))


class AccelRLEval(AccelRLBase):
    """Runs RL; tracks learning performance offline using evaluation trajectories
    (reference: accel_rl/runners/accel_rl.py:108-180).  Needs a sampler with
    `evaluate_policy(itr)` (GpuVecEvalSampler); the algorithm's optional `prep_eval(itr)` /
    `post_eval(itr)` hooks (dqn.py:209-216: epsilon switch) are called around it."""

    def __init__(self, eval_interval_steps=1e6, **kwargs):
        super().__init__(**kwargs)
        self._log_steps = int(eval_interval_steps)

    def train(self):
        n_itr = self.startup()
        for itr in range(n_itr):
            with logger.prefix("itr #%d | " % itr):
                if itr % self._log_interval_itrs == 0:
                    eval_traj_infos, eval_time = self.eval_policy(itr)
                    self.log_diagnostics(itr, eval_traj_infos, eval_time)
                samples_data, traj_infos = self.sampler.obtain_samples(itr)
                opt_data, opt_infos = self.algo.optimize_policy(itr, samples_data)
                self.store_diagnostics(itr, samples_data, opt_data, traj_infos, opt_infos)
        self.shutdown()

    def init_logging(self):
        self._cum_train_time = 0
        self._cum_eval_time = 0
        self._cum_total_time = 0
        super().init_logging()

    def eval_policy(self, itr):
        logger.log("evaluating policy...")
        start = time.time()
        getattr(self.algo, "prep_eval", lambda itr: None)(itr)
        traj_infos = self.sampler.evaluate_policy(itr)
        getattr(self.algo, "post_eval", lambda itr: None)(itr)
        logger.log("evaluation run complete")
        return traj_infos, time.time() - start

    def store_diagnostics(self, itr, samples_data, opt_data, traj_infos, opt_infos):
        for k, v in opt_infos.items():
            self._opt_infos[k].extend(v if isinstance(v, list) else [v])

    def log_diagnostics(self, itr, eval_traj_infos, eval_time):
        """reference: accel_rl.py:150-180"""
        self.save_itr_snapshot(itr)
        if not eval_traj_infos:
            logger.log("ERROR: had no complete trajectories in eval.")
        logger.record_tabular("Iteration", itr)
        logger.record_tabular("CumCompletedSteps", itr * self._sample_size)
        logger.record_tabular("StepsInEval", sum(info["Length"] for info in eval_traj_infos))
        logger.record_tabular("TrajsInEval", len(eval_traj_infos))
        self._log_infos(eval_traj_infos)
        if torch.device(self.policy.device).type == "cuda":
            torch.cuda.synchronize(self.policy.device)
        new_time = time.time()
        log_interval_time = new_time - self._last_time
        new_train_time = log_interval_time - eval_time
        self._cum_train_time += new_train_time
        self._cum_eval_time += eval_time
        self._cum_total_time += log_interval_time
        self._last_time = new_time
        train_speed = float("nan") if itr == 0 else self._log_interval_itrs * self._sample_size / new_train_time
        logger.record_tabular("CumTrainTime", self._cum_train_time)
        logger.record_tabular("CumEvalTime", self._cum_eval_time)
        logger.record_tabular("CumTotalTime", self._cum_total_time)
        logger.record_tabular("SamplesPerSecond", train_speed)
        self.last_tabular = logger.dump_tabular(with_prefix=False)
        logger.log("optimizing over {} iterations".format(self._log_interval_itrs))
