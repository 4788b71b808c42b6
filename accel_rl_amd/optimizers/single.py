"""Single-GPU optimizers (reference: accel_rl/optimizers/single/a2c_optimizer.py:11-50,
single/ppo_optimizer.py:11-75)."""
import numpy as np
import torch

from accel_rl_amd import _lib

from accel_rl_amd.optimizers.base import BaseOptimizer, iterate_mb_idxs


class A2cOptimizer(BaseOptimizer):
    """One gradient step on the whole batch."""

    def __init__(self, learning_rate, update_method, update_method_args=None, grad_norm_clip=None):
        self._learning_rate = learning_rate
        self._update_method = update_method
        self._update_args = update_method.resolve(**(update_method_args or dict()))
        self._grad_norm_clip = grad_norm_clip

    def initialize(self, inputs, losses, constraints, target, givens=None, lr_mult=1):
        self._input_names = list(inputs)
        self._losses = losses
        self._setup_bucket(target, lr_mult, givens)

    def optimize(self, inputs):
        self.prepare_host(len(inputs[0]))
        return self.device_updates(inputs)

    def prepare_host(self, data_length):
        self._set_updates_per_call(1)

    def device_updates(self, inputs):
        minibatch = dict(zip(self._input_names, inputs))
        minibatch["idx"] = None
        loss = self._backward(self._losses, minibatch)
        self._share_grad()
        self._apply_update(self._avg_factor())
        return loss, self._recent_grad_norms(1)

    def _share_grad(self):
        pass

    def _avg_factor(self):
        return 1.0

    @property
    def parallelism_tag(self):
        return "single"


class PpoOptimizer(BaseOptimizer):
    """epochs x minibatches over a batch that stays resident in HBM; minibatch rows
    are selected by host-shuffled index vectors (np.random.shuffle, one
    permutation per epoch, tail dropped -- optimizers/util.py:8-18)."""

    def __init__(self, learning_rate, update_method, update_method_args, epochs, minibatch_size,
                 grad_norm_clip=None, shuffle=True, num_slices=1):
        self._learning_rate = learning_rate
        self._update_method = update_method
        self._update_args = update_method.resolve(**(update_method_args or dict()))
        self._epochs = epochs
        self._minibatch_size = minibatch_size
        self._shuffle = shuffle
        self._grad_norm_clip = grad_norm_clip

    def initialize(self, inputs, losses, constraints, target, givens=None, lr_mult=1):
        self._input_names = list(inputs)
        self._losses = losses
        self._setup_bucket(target, lr_mult, givens)
        self._idx_host = None

    def optimize(self, inputs):
        self.prepare_host(len(inputs[0]))
        return self.device_updates(inputs)

    # Split for hipGraph capture: everything that draws from the host RNG happens in
    # prepare_host(); device_updates() only enqueues static-shape device work.
    def prepare_host(self, data_length):
        """Minibatch permutations for ALL epochs up front, into one pinned buffer: same
        RNG consumption order as the reference (np.random.shuffle once per epoch,
        optimizers/util.py:10-11; nothing else draws inside optimize)."""
        bs = self._minibatch_size
        flat = [mb for _ in range(self._epochs)
                for mb in iterate_mb_idxs(bs, data_length, self._shuffle)]
        if self._idx_host is None:
            shape = (max(len(flat), 1), bs)
            self._idx_host = torch.zeros(shape, dtype=torch.int32).pin_memory()
            self._idx_dev = torch.zeros(shape, dtype=torch.int32, device=self._target.device)
        self._n_minibatches = len(flat)
        self._set_updates_per_call(len(flat))
        if flat:
            self._idx_host.copy_(torch.from_numpy(np.stack(flat).astype(np.int32)))

    def device_updates(self, inputs):
        data = dict(zip(self._input_names, inputs))      # "_f_load": already on the device
        if not self._n_minibatches:
            return [], []
        _lib.copy_bytes(self._idx_dev, self._idx_host)
        if self._overlap_allreduce:
            return self._overlapped_minibatches(data)
        losses = []
        corun = self._corun_hook() if self.parallelism_tag == "single" else None
        try:
            for k in range(self._n_minibatches):
                mb = self._minibatch(data, self._idx_dev[k])
                if corun is not None:
                    mb["dense_w_hook"] = corun
                losses.append(self._backward(self._losses, mb))
                self._share_grad()
                self._apply_update(self._avg_factor())
        finally:
            self._hole = None           # (a backward pass that raised leaves no hole behind for the next call)
        return losses, self._recent_grad_norms(self._n_minibatches)

    def _minibatch(self, data, idx):
        return dict(data, idx=idx)               # the kernels gather the minibatch's rows by idx themselves

    # Multi-GPU (`_overlap_allreduce`, set by the sync optimizers): minibatches are enqueued eagerly -- at ~550 us
    # of device work per minibatch the host stays far ahead, and a hipGraph per minibatch only added its launch
    # latency at every hand-over to RCCL's stream (measured at world size 1: 232 k -> 237 k env-steps/s without
    # the graphs).  The policy reports, through `split_hook`, the point of its backward pass from which the tail
    # of the gradient bucket (dense layers + heads: 98 % of the bytes for spec 1) is final: that tail's
    # all-reduce is issued there and runs on RCCL's stream underneath the conv layers' backward (~60 % of the
    # minibatch); the small head of the bucket follows, and the optimiser step waits for both.
    _overlap_allreduce = False

    def _overlapped_minibatches(self, data):
        losses = []
        for k in range(self._n_minibatches):
            pending = []
            mb = self._minibatch(data, self._idx_dev[k])
            mb["split_hook"] = lambda: pending.append(self._share_grad_async(tail=True))
            losses.append(self._backward(self._losses, mb))
            pending.append(self._share_grad_async(tail=False) if pending else self._share_grad_async(None))
            self._wait_all(pending)
            self._apply_update(self._avg_factor())
        return losses, self._recent_grad_norms(self._n_minibatches)

    @staticmethod
    def _wait_all(pending):
        for work in pending:
            if work is not None:
                work.wait()

    def _share_grad_async(self, tail):
        """All-reduce of the gradient bucket's tail (True), head (False) or all of it (None); a Work or None."""
        return None

    def _share_grad(self):
        pass

    def _avg_factor(self):
        return 1.0

    @property
    def parallelism_tag(self):
        return "single"
