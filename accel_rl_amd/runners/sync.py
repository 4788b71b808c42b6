"""Synchronous multi-GPU runner: one process per GPU, every rank runs its own
sampler shard + learner, gradients are all-reduced inside the optimizer.

Reference: accel_rl/runners/multigpu_rl_base.py:10-153,216-231 and the MRO
composition accel_rl/runners/multigpu_rl.py:7-15 (AccelRLSync / SyncWorker).
There the master forks n-1 worker runners and bootstraps an NCCL clique through
a multiprocessing Manager dict; here ranks are separate processes either forked by rank 0's
`startup()` from the list of affinities (as the reference does) or started by
`python -m torch.distributed.run` (one per GPU); rendez-vous is the
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
    """`affinities` is the list of per-GPU dicts of the reference (multigpu_rl_base.py:20-45), one runner per entry.
    Two ways to get the n processes, one per GPU:

    * started as a plain `python script.py` (no WORLD_SIZE in the environment): `train()` / `startup()` forks the
      n - 1 worker runners itself, as the reference's `launch_workers` does -- same point in the program (nothing of
      the sampler, policy or algorithm is initialised yet, so the children inherit what the script built), same
      seeds (`seed + 100 * rank`), rank 0 stays in the calling process and joins the workers at shutdown;
    * started under `python -m torch.distributed.run` (RANK / WORLD_SIZE set): every process runs the script and is
      its own rank; a list of affinities must then have exactly WORLD_SIZE entries (ValueError otherwise).

    A single dict (or None) means one runner."""

    def __init__(self, affinities=None, seed=None, backend=None, **kwargs):
        self._backend = backend
        self._worker_affinities = None          # set: this process still has to fork ranks 1 .. n-1 (launch_workers)
        self._worker_procs = []
        as_list = list(affinities) if isinstance(affinities, (list, tuple)) else None
        if "WORLD_SIZE" in os.environ:                                # launched: one process per rank already exists
            self.rank = int(os.environ.get("RANK", 0))
            self.n_runners = int(os.environ["WORLD_SIZE"])
            if as_list is not None:
                if len(as_list) != self.n_runners:
                    raise ValueError("AccelRLSync got %d affinities but WORLD_SIZE=%d: one entry per launched rank "
                                     "(multigpu_rl_base.py:21)" % (len(as_list), self.n_runners))
                affinities = as_list[self.rank]
            if seed is not None:
                seed = seed + 100 * self.rank                          # multigpu_rl_base.py:28
        else:
            self.rank = 0
            self.n_runners = len(as_list) if as_list is not None else 1
            if as_list is not None:
                if not as_list:
                    raise ValueError("AccelRLSync: empty list of affinities")
                affinities = as_list[0]
                if self.n_runners > 1:
                    self._worker_affinities = as_list
        if affinities is None:
            affinities = dict(gpu=int(os.environ.get("LOCAL_RANK", self.rank)))
        super().__init__(affinities=affinities, seed=seed, **kwargs)

    # ------------------------------------------------------------------ ranks
    def launch_workers(self):
        """reference: MultiGpuRLBase.launch_workers (multigpu_rl_base.py:20-45): n_runners = len(affinities); workers
        1 .. n-1 get seed + 100 * rank and affinities[rank] and are forked here, before anything touches the GPU.  The
        Manager dict / barrier / queue of the reference are torch.distributed's TCP store on 127.0.0.1."""
        import multiprocessing as mp
        import threading
        from accel_rl_amd.util.misc import make_seed
        # (torch.cuda.is_initialized() only: is_available() / device_count() themselves mark the process so that a forked
        # child refuses the GPU -- "Cannot re-initialize CUDA in forked subprocess")
        if torch.cuda.is_initialized():
            raise RuntimeError("AccelRLSync cannot fork its %d worker runners: this process has already initialised the "
                               "GPU runtime (a forked child cannot use it).  Construct and train() the runner before any "
                               "device work, or start the script under `python -m torch.distributed.run "
                               "--nproc-per-node %d`." % (self.n_runners - 1, self.n_runners))
        table, self._worker_affinities = self._worker_affinities, None
        self._launched_here = True
        if self.seed is None:
            self.seed = make_seed()                                    # :22-23 (workers derive theirs from it)
        # Rendez-vous without touching os.environ (a second runner built later in this process must not find a stale
        # WORLD_SIZE and take the launched-by-torchrun road) and without a bind / close / re-bind window on the port:
        # rank 0 creates the TCP store on port 0 AFTER the fork and hands the port it got to the workers through shared
        # memory (startup()).
        ctx = mp.get_context("fork")
        self._rendezvous = (ctx.Array("i", 1), ctx.Event())
        base_seed = self.seed
        for rank in range(1, self.n_runners):
            p = ctx.Process(target=self._worker_main, args=(rank, base_seed + 100 * rank, table[rank]), daemon=True)
            p.start()
            self._worker_procs.append(p)

        def monitor(procs=tuple(self._worker_procs)):
            # a worker that dies leaves rank 0 inside a collective that never completes: say so and end the job
            while True:
                time.sleep(1.0)
                if getattr(self, "_workers_joined", False):
                    return
                dead = [(i + 1, p.exitcode) for i, p in enumerate(procs) if p.exitcode not in (None, 0)]
                if dead:
                    import sys
                    sys.stderr.write("AccelRLSync: worker runner(s) %s exited abnormally; ending the job\n" % dead)
                    sys.stderr.flush()
                    for p in procs:
                        if p.is_alive():
                            p.terminate()
                    self._abandon_sampler()
                    os._exit(70)
        threading.Thread(target=monitor, daemon=True).start()

    def _worker_main(self, rank, seed, affinities):
        """A forked worker runner (the reference's SyncWorker.train, multigpu_rl_base.py:66-103): same loop, own rank."""
        code = 0
        try:
            self.rank, self.seed, self.affinities = rank, seed, affinities
            self._worker_procs, self._launched_here = [], True
            if self.affinities is None:
                self.affinities = dict(gpu=rank)
            self.train()
        except BaseException as e:     # noqa: BLE001 -- the exit code is the message to rank 0's monitor
            import traceback
            traceback.print_exc()
            if "forked subprocess" in str(e):
                import sys
                sys.stderr.write("AccelRLSync: the script queried the GPU runtime (torch.cuda.is_available(), device_count(), a "
                                 "device tensor ...) before train() forked its worker runners; move that after train() starts or "
                                 "launch the script under `python -m torch.distributed.run`\n")
            code = 1
            self._abandon_sampler()
        finally:
            import sys
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(code)             # no interpreter finalisation in a forked child (the parent's atexit hooks are not ours)

    def _abandon_sampler(self):
        """Failure paths leave through os._exit (no atexit, no multiprocessing reaping): a sampler with processes of its
        own (HostEnvSampler's simulation workers) must be told to end them first, best effort and bounded."""
        try:
            kill = getattr(self.sampler, "kill_workers", None)
            if kill is not None:
                kill()
        except Exception:       # noqa: BLE001
            pass

    def shutdown(self):
        """reference: SyncBase.shutdown (multigpu_rl_base.py:124-127): rank 0 joins its workers; a worker only closes its
        sampler (MultiGpuWorkerBase.shutdown, :92-93).  A process group this runner created for ranks it forked itself is
        also torn down here (under a launcher that belongs to the script)."""
        if self.rank == 0:
            super().shutdown()
        else:
            self.sampler.shutdown()
        if self._launched_here and self.n_runners > 1 and dist.is_initialized():
            import threading
            dist.barrier()
            t = threading.Timer(20.0, lambda: os._exit(0))         # a teardown that blocks must not hang a finished run
            t.daemon = True
            t.start()
            dist.destroy_process_group()
            t.cancel()
        for p in self._worker_procs:
            p.join(60)
        self._workers_joined = True

    _launched_here = False      # True in the process that forked its workers and in those workers
    _rendezvous = None          # (port box, ready event) shared with the workers this runner forked

    def startup(self):
        """reference: multigpu_rl_base.py:12-18 (master) / :85-91 (worker)"""
        if self._worker_affinities is not None:
            self.launch_workers()
        if self.n_runners > 1 and not dist.is_initialized():
            backend = self._backend or ("nccl" if torch.cuda.is_available() else "gloo")     # (after the fork)
            kw = dict()
            if backend == "nccl":
                # accel_rl_base.py:62-64: the runner's GPU is affinities["gpu"]; RCCL binds its communicator to it
                gpu = self.affinities.get("gpu") if hasattr(self.affinities, "get") else None
                local = self.rank if self._rendezvous is not None else int(os.environ.get("LOCAL_RANK", self.rank))
                dev = torch.device("cuda", local if gpu is None else int(gpu))
                torch.cuda.set_device(dev)
                kw["device_id"] = dev
            if self._rendezvous is not None:                   # ranks this runner forked itself (launch_workers)
                import datetime
                box, ready = self._rendezvous
                if self.rank == 0:
                    store = dist.TCPStore("127.0.0.1", 0, self.n_runners, is_master=True, wait_for_workers=False,
                                          timeout=datetime.timedelta(seconds=300))
                    box[0] = store.port
                    ready.set()
                else:
                    if not ready.wait(300):
                        raise RuntimeError("AccelRLSync worker %d: rank 0 never opened the rendez-vous store" % self.rank)
                    store = dist.TCPStore("127.0.0.1", int(box[0]), self.n_runners, is_master=False,
                                          timeout=datetime.timedelta(seconds=300))
                kw["store"] = store
            dist.init_process_group(backend, rank=self.rank, world_size=self.n_runners, **kw)
        if self.rank != 0:
            logger.set_quiet(True)
        n_itr = super().startup(master=True)
        self.init_comm()
        self._start_time = self._last_time = time.time()
        return n_itr

    def init_comm(self):
        """reference: SyncBase.init_comm / SyncWorkerBase.init_comm (:111-122,136-149)"""
        if self.n_runners > 1:
            # one flat gradient bucket per all-reduce: every runner of the clique must have built the same network
            # (the reference forks ONE policy object, multigpu_rl_base.py:24-33; different games per rank need one
            # action space -- SynthAtariEnv(pad_actions_to=...))
            n = torch.tensor([self.policy.flat_params.numel()], dtype=torch.int64, device=self.policy.flat_params.device)
            lo, hi = n.clone(), n.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            if int(lo) != int(hi):
                raise ValueError("AccelRLSync: the runners' policies differ in size (%d ... %d parameters, %d here): a "
                                 "synchronous clique all-reduces ONE flat bucket" % (int(lo), int(hi), int(n)))
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
