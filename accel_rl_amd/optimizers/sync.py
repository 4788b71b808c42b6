"""Synchronous multi-GPU optimizers: local gradient -> all-reduce(SUM) of the ONE
flat fp32 gradient bucket over RCCL/xGMI -> x 1/n_gpu -> global-norm clip ->
identical local update on every rank.

Reference: accel_rl/optimizers/sync/base.py:10-24 (init_comm, _share_grad),
sync/sync_a2c_optimizer.py:13-59, sync/sync_ppo_optimizer.py:13-78; order
"avg, then norm/clip" from sync_ppo_optimizer.py:27-34 / optimizers/util.py:63-67.
The collective is torch.distributed (backend "nccl" == RCCL on ROCm; "gloo" for
the CPU tests of the protocol)."""
import torch
import torch.distributed as dist

from accel_rl_amd.optimizers.single import A2cOptimizer, PpoOptimizer


class _SyncMixin(object):
    _comm = None
    _n_gpu = 1
    _rank = 0
    _overlap_allreduce = True   # PpoOptimizer._overlapped_minibatches: bucket tail all-reduced under the conv backward
    _force_collective = False   # issue the all-reduce even with one rank (plumbing tests)
    # One hipGraph for the whole optimize_policy call, RCCL's all-reduces captured inside (aac_base._enqueue_optimize):
    # the eager path pays a compute -> RCCL stream -> compute hand-over per all-reduce on the host's launch path; inside a
    # graph they are edges.  Only on the `nccl` backend (gloo copies through the host: not capturable).
    graph_collectives = True
    # measurement hook (bench.py, N > 1, AFTER the timed region): the same learner with its collectives left out, so
    # that the line can state how much of the step the all-reduces add (`allreduce_exposed_ms`).  Every rank's
    # parameters then follow its own gradient: never set on a run whose result is used.
    _elide_collective = False

    def graph_ready(self):
        """True when this optimizer's collectives can be captured into a hipGraph."""
        if not self.graph_collectives:
            return False
        if not (self._n_gpu > 1 or self._force_collective):
            return True                     # no collective is issued at all
        return dist.is_initialized() and dist.get_backend(self._comm) == "nccl"

    def ranks_agree(self, ok):
        """True iff `ok` holds on every rank (one tiny eager all-reduce; no collective when there is one rank)."""
        if not (self._n_gpu > 1 and dist.is_initialized()):
            return bool(ok)
        flag = torch.tensor([0 if ok else 1], dtype=torch.int32, device=self._target.flat_grads.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self._comm)
        return int(flag.item()) == 0

    def init_comm(self, gpu_comm, rank, n_gpu):
        """`gpu_comm`: a torch.distributed process group (None = default group)."""
        self._comm = gpu_comm
        self._rank = rank
        self._n_gpu = n_gpu

    # test hook: a list that receives (first element, clone of the LOCAL gradient slice) right before each all-reduce
    # -- what this rank contributes to the sum (tests/test_sync_gpu.py pins the update to the mean of the ranks' taps)
    _grad_tap = None

    def _tap(self, g, first):
        if self._grad_tap is not None:
            self._grad_tap.append((int(first), g.detach().clone()))

    def _share_grad(self):
        if (self._n_gpu > 1 or self._force_collective) and not self._elide_collective:
            self._tap(self._target.flat_grads, 0)
            dist.all_reduce(self._target.flat_grads, op=dist.ReduceOp.SUM, group=self._comm)

    def _share_grad_async(self, tail):
        """The same all-reduce, asynchronous, of the bucket's tail / head / whole (the split point is the
        policy's `grad_split_offset`: everything behind it is final when the policy calls the split hook)."""
        if not (self._n_gpu > 1 or self._force_collective) or self._elide_collective:
            return None
        g, first = self._target.flat_grads, 0
        if tail is not None:
            off = int(getattr(self._target, "grad_split_offset", 0))
            g, first = (g[off:], off) if tail else (g[:off], 0)
            if g.numel() == 0:
                return None
        self._tap(g, first)
        return dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self._comm, async_op=True)

    def _avg_factor(self):
        return 1.0 / self._n_gpu

    @property
    def parallelism_tag(self):
        return "synchronous"


class SyncA2cOptimizer(_SyncMixin, A2cOptimizer):
    pass


class SyncPpoOptimizer(_SyncMixin, PpoOptimizer):
    pass
