"""Synchronous multi-GPU runner: one process per GPU, every rank runs its own
sampler shard + learner, gradients are all-reduced inside the optimizer.

Reference: accel_rl/runners/multigpu_rl_base.py:10-153,216-231 and the MRO
composition accel_rl/runners/multigpu_rl.py:7-15 (AccelRLSync / SyncWorker).
There the master forks n-1 worker runners and bootstraps an NCCL clique through
a multiprocessing Manager dict; here ranks are separate processes started by
`python -m torch.distributed.run` (one per GPU), rendez-vous is the
torch.distributed TCP store, the initial parameters are broadcast from rank 0
(replaces the pickled vector in the manager dict, multigpu_rl_base.py:119,
142-143) and completed-trajectory infos are gathered to rank 0 at log time
(replaces the mp.Queue, :216-231).  Per-rank seed = seed + 100 * rank (:28);
n_itr counts sample_size * n_runners per iteration (:62-63).
"""
import os
import time

import torch
import torch.distributed as dist

from accel_rl_amd.runners.accel_rl import AccelRL
from accel_rl_amd.util import logger


class AccelRLSync(AccelRL):

    def __init__(self, affinities=None, seed=None, backend=None, **kwargs):
        self._backend = backend
        self.rank = int(os.environ.get("RANK", 0))
        self.n_runners = int(os.environ.get("WORLD_SIZE", 1))
        if isinstance(affinities, (list, tuple)):                  # list of per-GPU dicts
            affinities = affinities[self.rank]
        if affinities is None:
            affinities = dict(gpu=int(os.environ.get("LOCAL_RANK", self.rank)))
        if seed is not None:
            seed = seed + 100 * self.rank                          # multigpu_rl_base.py:28
        super().__init__(affinities=affinities, seed=seed, **kwargs)

    def startup(self):
        """reference: multigpu_rl_base.py:12-18 (master) / :85-91 (worker)"""
        if self.n_runners > 1 and not dist.is_initialized():
            dist.init_process_group(self._backend or ("nccl" if torch.cuda.is_available() else "gloo"))
        if self.rank != 0:
            logger.set_quiet(True)
        n_itr = super().startup(master=True)
        self.init_comm()
        self._start_time = self._last_time = time.time()
        return n_itr

    def init_comm(self):
        """reference: SyncBase.init_comm / SyncWorkerBase.init_comm (:111-122,136-149)"""
        if self.n_runners > 1:
            dist.broadcast(self.policy.flat_params, src=0)          # initial_param_values
            self._initial_param_vector = self.policy.flat_params.clone()
        self.algo.optimizer.init_comm(None, self.rank, self.n_runners)
        if self.n_runners > 1:
            dist.barrier()

    def get_n_itr(self, sample_size):
        n_itr = super().get_n_itr(sample_size * self.n_runners)    # :62-63
        return n_itr

    def store_diagnostics(self, itr, samples_data, opt_data, traj_infos, opt_infos):
        """OnlineLog / OnlineLogWorker (:216-231): rank 0 also sees the workers' episodes."""
        if self.n_runners > 1 and (itr + 1) % self._log_interval_itrs == 0:
            pending = getattr(self, "_pending_traj", []) + list(traj_infos)
            gathered = [None] * self.n_runners
            dist.all_gather_object(gathered, [dict(t) for t in pending])
            self._pending_traj = []
            traj_infos = [t for part in gathered for t in part] if self.rank == 0 else []
        elif self.n_runners > 1:
            self._pending_traj = getattr(self, "_pending_traj", []) + list(traj_infos)
            traj_infos = []
        super().store_diagnostics(itr, samples_data, opt_data, traj_infos, opt_infos)

    def log_diagnostics(self, itr):
        if self.rank == 0:
            super().log_diagnostics(itr)
        else:
            self._opt_infos = {k: list() for k in self._opt_infos}

    @property
    def parallelism_tag(self):
        return "synchronous"
