"""DQN-family optimizer (reference: accel_rl/optimizers/single/dqn_optimizer.py:11-54): one gradient
step per call on a replay minibatch, optional conv-gradient scaling and global-norm clip, adam / rmsprop
on the flat bucket (csrc/optim.hip).  `loss` is a callable(minibatch tuple) -> (priority f32[B], loss scalar) that leaves
the gradient in the policy's flat bucket (the algorithm builds it)."""
import numpy as np
import torch

from accel_rl_amd.optimizers.base import BaseOptimizer
from accel_rl_amd.util.misc import graph_capture_mode


class DqnOptimizer(BaseOptimizer):
    """use_graph: after two eager calls the whole update (three forward passes, loss, backward, optimiser
    step: ~60 launches) is replayed from one hipGraph over static input buffers -- at the reference's
    batch size of 32 the update is launch-bound, not compute-bound."""

    def __init__(self, learning_rate, update_method, update_method_args=None, grad_norm_clip=None,
                 scale_conv_grads=False, use_graph=True):
        self._scale_conv_grads = scale_conv_grads      # dueling networks: conv gradients x 2^-1/2 (:34-36)
        self._learning_rate = learning_rate
        self._update_method = update_method
        self._update_args = update_method.resolve(**(update_method_args or dict()))
        self._grad_norm_clip = grad_norm_clip
        self._use_graph = use_graph
        self._static = self._graph = self._graph_out = None
        self._ring, self._pack_rows = None, 0
        self._calls = 0

    def initialize(self, inputs, loss, target, priority_expr=None, givens=None, lr_mult=1):
        self._input_names = list(inputs)
        self._loss_fn = loss
        self._setup_bucket(target, lr_mult)
        self._set_updates_per_call(1)

    def optimize(self, inputs):
        if not self._use_graph:
            return self._step(inputs)
        static = self._stage(inputs)
        self._calls += 1
        if self._graph is None:
            if self._calls <= 2:                                  # warm-up: buffers, kernel attributes
                return self._step(static)
            if self._pack_rows:                                   # (allocated here: not from the capture's private pool)
                dev = self._target.device
                self._ring = torch.zeros((self.STATS_RING, 2 * self._pack_rows), dtype=torch.float32, device=dev)
                self._ring_count = torch.zeros(1, dtype=torch.int32, device=dev)
                self._ring_replays = 0
            torch.cuda.synchronize(self._target.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode=graph_capture_mode()):
                self._graph_out = self._step(static)
            self._graph = graph
        self._graph.replay()
        priority, loss = self._graph_out          # graph-owned: valid until the next replay
        if self._ring is not None and loss.dim() == 1:     # the graph left [loss rows | priorities] in a ring slot: no launch here
            slot = self._ring_replays % self._ring.shape[0]
            self._ring_replays += 1
            return priority, self._ring[slot]
        return priority, loss.clone()

    STATS_RING = 1024           # updates whose statistics stay readable (one optimize_policy call enqueues at most this many)

    def _step(self, inputs):
        priority, loss = self._loss_fn(inputs)
        if loss.dim() == 1:                            # per-row losses (the loss is their sum)
            b = loss.numel()
            capturing = loss.is_cuda and torch.cuda.is_current_stream_capturing()
            packed = (priority.dim() == 1 and priority.numel() == b and loss.is_contiguous() and priority.is_contiguous()
                      and priority.data_ptr() == loss.data_ptr() + 4 * b)
            self._pack_rows = b if packed else 0           # (the warm-up calls tell optimize() how wide the ring is)
            if capturing and packed and self._ring is not None and self._ring.shape[1] == 2 * b:
                # inside the graph the statistics of an update are ONE ring append of both rows (the loss's sum and the
                # downsampled priorities are taken from the ring after the update loop: algos/dqn/dqn.py)
                from accel_rl_amd import _lib
                _lib.ring_append(torch.as_strided(loss, (2 * b,), (1,)), self._ring, self._ring_count)
            else:
                loss = loss.sum()
        if self._scale_conv_grads:                     # the conv tensors lead the bucket (optimizers/util.py:122-126)
            self._target.flat_grads[:self._target.grad_split_offset].mul_(float(np.float32(2 ** -0.5)))
        self._apply_update(1.0)
        self._finish_updates(1)
        return priority, loss

    def _stage(self, inputs):
        """Copy one minibatch into the static buffers the graph reads (host arrays -- the importance
        weights -- through a pinned mirror guarded by an event)."""
        dev = self._target.device
        in_place = [isinstance(x, torch.Tensor) and getattr(x, "_arl_static", False) for x in inputs]
        if self._static is None or any(tuple(s.shape) != tuple(np.shape(x)) or (keep and s is not x)
                                       for s, x, keep in zip(self._static, inputs, in_place)):
            assert self._graph is None, "the replay minibatch changed shape or buffers after graph capture"
            self._static, self._pinned = [], []
            for x, keep in zip(inputs, in_place):
                if keep:                                          # the replay buffer's own static outputs
                    self._static.append(x)
                    self._pinned.append(None)
                elif isinstance(x, torch.Tensor):
                    self._static.append(torch.empty_like(x, device=dev))
                    self._pinned.append(None)
                else:
                    x = np.asarray(x, np.float32)
                    self._static.append(torch.empty(x.shape, dtype=torch.float32, device=dev))
                    self._pinned.append(torch.empty(x.shape, dtype=torch.float32).pin_memory())
            self._stage_event = torch.cuda.Event()
        else:
            self._stage_event.synchronize()                       # the previous upload has left the pinned mirrors
        copied = False
        for s, p, x in zip(self._static, self._pinned, inputs):
            if s is x:
                continue
            copied = True
            if p is None:
                s.copy_(x, non_blocking=True)
            else:
                p.copy_(torch.from_numpy(np.asarray(x, np.float32)))
                s.copy_(p, non_blocking=True)
        if copied:
            self._stage_event.record(torch.cuda.current_stream(dev))
        return tuple(self._static)

    @property
    def parallelism_tag(self):
        return "single"
