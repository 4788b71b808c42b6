"""Optimizer interface (reference: accel_rl/optimizers/base.py:3-13) and the
shared flat-bucket machinery.

`initialize(inputs, losses, constraints, target, givens=None, lr_mult=1)` keeps
the reference's argument names.  With Theano gone, `inputs` is the list of input
NAMES (tuple order of prep_opt_inputs, aac_base.py:147-170) and `losses` is a
callable `losses(minibatch: dict) -> loss4 = (pi_loss, v_loss, ent_loss, their sum)` that
runs the policy's HIP forward / backward and OVERWRITES the target's flat gradient
bucket; `target` is the policy; `lr_mult` a 1-element device tensor (the linear schedule
writes it)."""
import numpy as np
import torch

from accel_rl_amd import _lib

NORM_LOG_LEN = 64


def iterate_mb_idxs(batch_size, data_length, shuffle=False):
    """Minibatch index arrays; one fresh host permutation per call, tail dropped
    (reference: accel_rl/optimizers/util.py:8-18)."""
    if shuffle:
        order = np.arange(data_length)
        np.random.shuffle(order)
    for lo in range(0, data_length - batch_size + 1, batch_size):
        yield order[lo:lo + batch_size] if shuffle else np.arange(lo, lo + batch_size)


def iterate_traj_idxs(batch_size, data_length, horizon=1, shuffle=False):
    """Minibatches of whole segments for recurrent policies: yields (row indices of the chosen segments in
    time order, segment numbers); rows are segment-major (`horizon` consecutive rows per env), one host
    permutation of the segments per call (reference: accel_rl/optimizers/util.py:21-32)."""
    if data_length % horizon or batch_size % horizon or data_length % batch_size:
        raise AssertionError("batch_size and data_length must be multiples of horizon, data_length of batch_size")
    rows = np.arange(data_length).reshape(-1, horizon)
    order = np.arange(data_length // horizon)
    per = batch_size // horizon
    if shuffle:
        np.random.shuffle(order)
    for lo in range(0, len(order), per):
        chosen = order[lo:lo + per]
        yield rows[chosen].reshape(-1), chosen


class BaseOptimizer(object):

    def initialize(self, inputs, losses, constraints, target, givens=None, lr_mult=1):
        raise NotImplementedError

    def optimize(self, inputs):
        raise NotImplementedError

    @property
    def parallelism_tag(self):
        raise NotImplementedError

    # ---- shared: flat bucket + HIP update ------------------------------------
    def _setup_bucket(self, target, lr_mult, givens=None):
        self._target = target
        dev = target.device
        n = target.flat_params.numel()
        self._slot0 = torch.zeros(n, dtype=torch.float32, device=dev)
        self._slot1 = torch.zeros(n, dtype=torch.float32, device=dev) \
            if self._update_method.name == "adam" else None
        self._step_count = torch.zeros(1, dtype=torch.float32, device=dev)
        if not isinstance(lr_mult, torch.Tensor):
            lr_mult = torch.full((1,), float(lr_mult), dtype=torch.float32, device=dev)
        self._lr_mult = lr_mult
        self._partials = torch.zeros(_lib.OPT_PARTIALS, dtype=torch.float64, device=dev)
        self._norm_log = torch.zeros(NORM_LOG_LEN, dtype=torch.float32, device=dev)
        self._n_updates = 0
        self._step_pp = self._norm_parts = None      # the no-clip update's state (see _apply_update)
        self._hole = None                            # (first, count): range whose update this backward pass handed out
        self._call_hole = None                       # ... as fixed by the call's first update
        self._hole_count = 0                         # ... its length (0: plain updates)
        st = _lib.ArlOptState()
        st.n_params = n
        st.params, st.grads = target.flat_params.data_ptr(), target.flat_grads.data_ptr()
        st.slot0 = self._slot0.data_ptr()
        st.slot1 = self._slot1.data_ptr() if self._slot1 is not None else None
        st.step_count, st.lr_mult = self._step_count.data_ptr(), self._lr_mult.data_ptr()
        st.partials, st.grad_norm_log = self._partials.data_ptr(), self._norm_log.data_ptr()
        st.norm_log_len = NORM_LOG_LEN
        self._opt_state = st
        a = self._update_args
        if self._update_method.name == "adam":
            self._kernel_args = (a["beta1"], a["beta2"], a["epsilon"])
        else:
            self._kernel_args = (a["rho"], 0.0, a["epsilon"])

    def _backward(self, losses, minibatch):
        """Gradient of the summed loss into the flat gradient bucket (overwritten); returns the summed loss."""
        return losses(minibatch)[3]

    def _apply_update(self, avg_factor=1.0):
        """(avg) -> global norm (-> clip) -> adam/rmsprop.  With clipping: two HIP launches (the norm comes first).
        Without (PPO's default): ONE launch -- the norm is only logged, its sum of squares rides along in the
        update's pass over the gradient and `_recent_grad_norms` finishes the call's norms in one small launch."""
        b1, b2, eps = self._kernel_args
        if self._grad_norm_clip is None:
            self._noclip_state()
            k = self._n_updates % self._opt_state.norm_log_len      # position inside the call
            hole, self._hole = self._hole, None
            if k == 0:                               # the first update decides how the whole call is split
                self._call_hole = hole
                self._hole_count = hole[1] if hole else 0
            elif self._hole_count and hole is None:  # a later backward pass did not hand its hole over: run it here
                hole = self._call_hole
                _lib.corun_job_run(_lib.corun_job(self._opt_state, self._update_method.kernel_id, self._learning_rate,
                                                  avg_factor, b1, b2, eps, k, self._step_pp, self._norm_parts,
                                                  hole[0], hole[1]))
            if self._hole_count:
                hole = self._call_hole
                _lib.opt_step_noclip_split(self._opt_state, self._update_method.kernel_id, self._learning_rate,
                                           avg_factor, b1, b2, eps, k, self._step_pp, self._norm_parts, hole[0],
                                           hole[1], 0)
            else:
                _lib.opt_step_noclip(self._opt_state, self._update_method.kernel_id, self._learning_rate, avg_factor,
                                     b1, b2, eps, k, self._step_pp, self._norm_parts)
            self._pending_avg = avg_factor
        else:
            _lib.opt_step(self._opt_state, self._update_method.kernel_id, self._learning_rate,
                          avg_factor, self._grad_norm_clip, b1, b2, eps)
        self._n_updates += 1

    corun_update = True             # (tests switch it off to compare with the plain one-launch update)

    def _noclip_state(self):
        if self._step_pp is None:
            dev = self._target.device
            self._step_pp = self._step_count.repeat(2).contiguous()
            self._norm_parts = torch.zeros(_lib.OPT_NORM_SLOTS * _lib.OPT_NORM_BLOCKS, dtype=torch.float64, device=dev)
        elif self._n_updates % self._opt_state.norm_log_len == 0 and not self._hole:
            # start of a call: Lasagne's t as it stands in step_count (an external write -- a restored checkpoint --
            # since the last call must not be overwritten by this call's opt_finish)
            self._step_pp.copy_(self._step_count.expand(2))

    def _corun_hook(self, avg_factor=1.0):
        """For the policy's `dense_w_hook`: without norm clipping nothing has to wait for the whole gradient, so the
        update of a range that is final early -- the first dense layer's weights, 98 % of the bucket's bytes for
        spec 1 -- rides inside the next data-gradient launch (HBM-bound streaming in extra workgroups of an
        MFMA-bound kernel: measured free) and `_apply_update` only updates the rest.  Same arithmetic per element."""
        if self._grad_norm_clip is not None or not self.corun_update:
            return None
        b1, b2, eps = self._kernel_args

        def hook(first, count):
            """-> the job (plain data: nothing is pending anywhere if the caller drops it), or None when this call's
            updates are not split (its first backward pass offered no hole) or are split elsewhere."""
            self._noclip_state()
            k = self._n_updates % self._opt_state.norm_log_len
            if k and (self._hole_count == 0 or self._call_hole != (first, count)):
                return None
            self._hole = (first, count)
            return _lib.corun_job(self._opt_state, self._update_method.kernel_id, self._learning_rate, avg_factor,
                                  b1, b2, eps, k, self._step_pp, self._norm_parts, first, count)
        return hook

    def _set_updates_per_call(self, count):
        """The update kernel logs the grad norm of update number t at slot (t-1) % len.
        With len == updates per optimize() call, update k of EVERY call lands on slot k,
        so a captured hipGraph can read fixed slots."""
        count = max(1, min(int(count), NORM_LOG_LEN))
        if self._opt_state.norm_log_len != count:
            assert self._n_updates % count == 0 or self._n_updates == 0
            self._opt_state.norm_log_len = count

    def _recent_grad_norms(self, count):
        """Device tensor [count]: global grad norms of this call's updates (no host sync)."""
        assert 0 < count <= self._opt_state.norm_log_len
        self._finish_updates(count)
        # (inside a graph capture the log slice itself is the output: it lives exactly as long as a clone made by the
        #  graph would -- until the next replay -- and the learner's graph copies it into its diagnostics ring anyway;
        #  eager callers keep the tensor across calls and get their own copy)
        out = self._norm_log[:count]
        return out if (out.is_cuda and torch.cuda.is_current_stream_capturing()) else out.clone()

    def _finish_updates(self, count):
        """Close a call of `count` no-clip updates (norm log, Lasagne's t); every call must end with this."""
        if self._grad_norm_clip is None and self._step_pp is not None:
            _lib.opt_finish(self._opt_state, count, self._pending_avg, self._step_pp, self._norm_parts,
                            hole_count=self._hole_count)
