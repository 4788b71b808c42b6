"""Advantage actor-critic base: process_samples on the device + loss definitions.

Reference: accel_rl/algos/pg/aac_base.py:15-177.  process_samples there is a
Python double loop over envs and time calling numpy scalar code
(algos/pg/util.py:6-63); here it is: one policy forward for the bootstrap values
-> arl_gae_scan or arl_nstep_return -> (arl_valids_mask) -> (arl_standardize),
all on the sampler's GPU, directly on the rollout buffer the sampler filled.
The loss graph (aac_base.py:60-70) is the policy's HIP forward / backward around the fused head kernel
(csrc/learner.hip: both heads, softmax, pi / value / entropy losses and their gradients in one pass).
"""
import torch

from accel_rl_amd import _lib
from accel_rl_amd.algos.base import RLAlgorithm
from accel_rl_amd.buffers import buffer_with_segs_view
from accel_rl_amd.util import logger
from accel_rl_amd.util.misc import graph_capture_mode
from accel_rl_amd.util.quick_args import save_args
import numpy as np

LR_SCHEDULES = ["linear"]


def valids_mean(expression, valids=None):
    """reference: algos/pg/util.py:49-53"""
    if valids is None:
        return expression.mean()
    v = valids.to(expression.dtype)
    return torch.sum(v * expression) * (1. / torch.sum(v))


class AdvActorCriticBase(RLAlgorithm):

    def __init__(self, discount, gae_lambda, v_loss_coeff=1, ent_loss_coeff=0.01,
                 standardize_adv=False, lr_schedule=None, promo="nep50", use_graph=True):
        if lr_schedule is not None and lr_schedule not in LR_SCHEDULES:
            raise ValueError("Unrecognized lr_schedule: {}, should be None (for constant) or "
                             "in: {}".format(lr_schedule, LR_SCHEDULES))
        save_args(vars(), underscore=False)
        self.need_extra_obs = True          # (signal sent to the sampler)
        # promo: numeric contract of the return / advantage scans (include/accel_rl_hip.h): "nep50" / "legacy" = the
        # reference's arithmetic bit for bit under numpy >= 2 / 1.x; "assoc" = wavefront suffix scan, within 1e-5
        self._promo = dict(nep50=_lib.PROMO_NEP50, legacy=_lib.PROMO_LEGACY, assoc=_lib.PROMO_ASSOC)[promo]

    def initialize(self, policy, env_spec, sample_size, horizon, mid_batch_reset):
        if mid_batch_reset and policy.recurrent:
            raise NotImplementedError
        dev = policy.device
        self.policy = policy
        self._lr_mult = torch.ones(1, dtype=torch.float32, device=dev)
        self._use_valids = not (mid_batch_reset and not policy.recurrent)   # aac_base.py:53-58
        self._dist_info_keys = policy.distribution.dist_info_keys
        input_names = ["observations", "actions", "advantages", "returns", "old_value"]
        input_names += ["old_%s" % k for k in self._dist_info_keys]
        self._state_info_keys = list(policy.state_info_keys)             # aac_base.py:45-46
        input_names += self._state_info_keys
        if self._state_info_keys and hasattr(self.optimizer, "prepare_host") and \
                getattr(self.optimizer, "_minibatch_size", None) is not None:
            # the reference's PpoOptimizer slices rows, not trajectories (ppo_optimizer.py:67-75): recurrent
            # policies are trained with the whole-batch optimizers only
            raise NotImplementedError("recurrent policies need a whole-batch optimizer (A2C) (INTEGRATION.md, section E)")
        opt_examples = dict(advantages=np.float32(1), returns=np.float32(1))
        if self._use_valids:
            input_names.append("valids")
            opt_examples["valids"] = np.int8(1)
        if not hasattr(policy, "loss_and_grads"):
            raise TypeError("the policy must provide loss_and_grads (HIP forward / backward into its flat "
                            "gradient bucket); got {}".format(type(policy).__name__))
        self.optimizer.initialize(inputs=input_names, losses=self._losses, constraints=None, target=policy,
                                  lr_mult=self._lr_mult)
        self._opt_buf = buffer_with_segs_view(opt_examples, sample_size, horizon, dev)
        self._batch_size = sample_size
        self._mid_batch_reset = mid_batch_reset
        self._horizon = horizon
        self._n_env = sample_size // horizon
        self._std_ws = _lib.standardize_workspace(dev)
        self._lr_mult_host = torch.ones(1, dtype=torch.float32).pin_memory()
        self._graph = None
        self._graph_out = None
        self._graph_samples = None
        self._warm_calls = 0
        self._done_event = None
        # diagnostics of a replayed update: appended INSIDE the graph to a device ring (arl_ring_append); the caller
        # gets slot (replay number % INFO_RING) -- valid for INFO_RING further iterations, no launch between two replays
        self._info_ring = None
        self._info_replays = 0

    def set_n_itr(self, n_itr):
        self.n_itr = n_itr

    def optimize_policy(self, itr, samples_data):
        """reference: aac_base.py:102-106.  Host-side draws (minibatch permutations, lr
        schedule) are made first; the device work (bootstrap forward, scan, epochs x
        minibatches of forward/backward/update) is static-shape and, after two eager
        warm-up calls, is replayed from one hipGraph."""
        # The pinned host buffers below (lr multiplier, minibatch indices) are read by copy nodes of the
        # PREVIOUS call's device work, which may still be queued (the sampler no longer waits for the
        # stream): wait for that call's own event -- not for the rollout already enqueued behind it.
        if self._done_event is not None:
            self._done_event.synchronize()
        if self.lr_schedule == "linear":                             # aac_base.py:165-168
            self._lr_mult_host.fill_(max((self.n_itr - itr) / self.n_itr, 0.))
        if hasattr(self.optimizer, "prepare_host"):
            self.optimizer.prepare_host(self._batch_size)
        out = self._enqueue_optimize(itr, samples_data)
        if self._done_event is None:
            self._done_event = torch.cuda.Event()
        self._done_event.record(torch.cuda.current_stream(self.policy.device))
        return out

    def _enqueue_optimize(self, itr, samples_data):
        tag = self.optimizer.parallelism_tag
        sync = tag == "synchronous" and getattr(self.optimizer, "graph_ready", lambda: False)()
        graphable = self.use_graph and hasattr(self.optimizer, "device_updates") and (tag == "single" or sync)
        if not graphable:
            return self._device_optimize(itr, samples_data)
        if self._graph is None:
            self._warm_calls += 1
            if self._warm_calls <= 2:
                out = self._device_optimize(itr, samples_data)
                self._ensure_info_ring(out[1])              # (allocated here: not from the capture's private pool)
                return out
            torch.cuda.synchronize(self.policy.device)
            graph, failure = torch.cuda.CUDAGraph(), None
            # a capture that fails part-way leaves the optimiser's host-side call counters advanced by a partial call
            # (no kernel ran): the eager re-run below must start from the state the capture started from
            opt_host = {k: getattr(self.optimizer, k) for k in
                        ("_n_updates", "_hole", "_call_hole", "_hole_count", "_pending_avg") if hasattr(self.optimizer, k)}
            try:
                with torch.cuda.graph(graph, capture_error_mode=graph_capture_mode()):
                    self._graph_out = self._device_optimize(itr, samples_data)
                    self._append_infos(self._graph_out[1])
            except Exception as e:          # single GPU: a bug, raise.  N > 1: every rank must take the same road
                if not sync:
                    raise
                failure = e
            if sync and not self.optimizer.ranks_agree(failure is None):
                # a rank could not capture its collectives: ALL ranks run the minibatches eagerly from here on (the
                # structure of round 2, same sums in the same order; tests/test_sync_gpu.py runs both)
                logger.log("WARNING: hipGraph capture of the synchronous learner failed on a rank (%r): eager "
                           "minibatches from here on" % (failure,))
                self.optimizer.graph_collectives = False
                for k, v in opt_host.items():
                    setattr(self.optimizer, k, v)
                torch.cuda.synchronize(self.policy.device)
                return self._device_optimize(itr, samples_data)
            self._graph, self._graph_samples = graph, samples_data
        assert samples_data is self._graph_samples, "the sampler must hand over the same buffer"
        self._graph.replay()
        # the graph owns its outputs and every replay overwrites them: callers keep the diagnostics across iterations
        # (AccelRL.store_diagnostics) -- the graph itself left this replay's copy in a ring slot
        opt_data, infos = self._graph_out
        # (the modulus is the ALLOCATED ring's length -- what ring_append's device counter wraps at -- not the current
        #  _ring_slots(): a log interval lowered after the ring was sized must not move the host's idea of the slot)
        slot = self._info_replays % next(iter(self._info_ring.values())).shape[0]
        self._info_replays += 1
        return opt_data, {k: self._info_ring[k][slot] for k in infos}

    INFO_RING = 4096            # least number of iterations a returned diagnostics tensor stays valid
    _log_interval_itrs = 0      # set_log_interval_itrs: how long the runner keeps the tensors before it reads them

    def set_log_interval_itrs(self, n):
        """The runner holds an iteration's diagnostics until its next log line (AccelRL.store_diagnostics: up to
        `log_interval_steps // sample_size` iterations, e.g. 12 500 for 80-step batches and 1e6-step intervals): the
        ring is sized so that no slot is rewritten while the runner still holds it."""
        self._log_interval_itrs = int(n)
        ring = getattr(self, "_info_ring", None)          # (allocated with the first replayed update)
        if ring is not None and self._ring_slots() > next(iter(ring.values())).shape[0]:
            raise RuntimeError("set_log_interval_itrs(%d) after the diagnostics ring was sized" % n)

    def _ring_slots(self):
        return max(self.INFO_RING, self._log_interval_itrs + 2)

    def _ensure_info_ring(self, infos):
        if self._info_ring is None:
            self._info_ring = {k: torch.zeros((self._ring_slots(),) + tuple(v.shape), dtype=torch.float32, device=v.device)
                               for k, v in infos.items()}
            self._info_counts = {k: torch.zeros(1, dtype=torch.int32, device=v.device) for k, v in infos.items()}
            self._info_replays = 0

    def _append_infos(self, infos):
        """(inside the capture) every diagnostics tensor of the call -> the next slot of its ring."""
        for k, v in infos.items():
            _lib.ring_append(v.reshape(-1), self._info_ring[k], self._info_counts[k])

    def _device_optimize(self, itr, samples_data):
        _lib.copy_bytes(self._lr_mult, self._lr_mult_host)
        opt_data = self.process_samples(itr, samples_data)
        opt_input_values = self.prep_opt_inputs(itr, samples_data, opt_data)
        if hasattr(self.optimizer, "device_updates"):
            _, grad_norm = self.optimizer.device_updates(opt_input_values)
        else:
            _, grad_norm = self.optimizer.optimize(opt_input_values)
        return opt_data, dict(GradNorm=grad_norm)

    def process_samples(self, itr, samples_data):
        """reference: aac_base.py:108-145"""
        n, t = self._n_env, self._horizon
        opt = self._opt_buf
        last_values = self.policy.value(samples_data["extra_observations"]).contiguous()   # :112
        r, d = samples_data["rewards"], samples_data["dones"]
        v = samples_data["agent_infos"]["value"]
        if self.gae_lambda == 1:                                    # :115-121
            _lib.nstep_return(r, d, v, last_values, self.discount, n, t,
                              opt["returns"], opt["advantages"], promo=self._promo)
        else:                                                        # :122-127
            _lib.gae_scan(r, v, d, last_values, self.discount, self.gae_lambda, n, t,
                          opt["advantages"], opt["returns"], promo=self._promo)
        valids = None
        if self._use_valids:                                         # :129-134
            flags = samples_data["env_infos"].get("need_reset", d)
            _lib.valids_mask(flags, n, t, opt["valids"], opt["advantages"], opt["returns"], v)
            valids = opt["valids"]
        if self.standardize_adv:                                     # :136-143
            _lib.standardize(opt["advantages"], valids, self._std_ws, 1e-6)
        return opt

    def prep_opt_inputs(self, itr, samples_data, opt_data):
        """reference: aac_base.py:147-170"""
        agent_infos = samples_data["agent_infos"]
        values = (samples_data["observations"], samples_data["actions"], opt_data["advantages"],
                  opt_data["returns"], agent_infos["value"])
        values += tuple(agent_infos[k] for k in self._dist_info_keys)
        # the stored previous hidden states; the policy reads only each segment's first row
        # (s[::horizon], aac_base.py:157-161)
        values += tuple(agent_infos[k] for k in self._state_info_keys)
        if self._use_valids:
            values += (opt_data["valids"],)
        return values

    # ---- loss graph (aac_base.py:60-70) ---------------------------------------
    def _losses(self, mb):
        """pi_loss + v_loss + ent_loss of one minibatch (aac_base.py:60-66, `pi_loss` of the subclass selected by
        `loss_kind`) and their gradient: the policy's forward, the fused head kernel and the backward pass write
        every gradient into flat_grads."""
        inv_count = None
        valids = mb.get("valids")
        if valids is not None:
            v = valids if mb.get("idx") is None else valids.index_select(0, mb["idx"].long())
            inv_count = (1. / v.sum(dtype=torch.float32)).reshape(1)
        if self._state_info_keys:
            mb = dict(mb, horizon=self._horizon)
        loss4 = self.policy.loss_and_grads(mb, self.loss_kind, getattr(self, "clip_param", 0.),
                                           self.v_loss_coeff, self.ent_loss_coeff, self._lr_mult,
                                           inv_count, tie_rule=self.loss_tie_rule)
        return loss4            # (pi_loss, v_loss, ent_loss, their sum): the optimizer reads [3]

    loss_kind = None        # 0 = A2C, 1 = PPO (selects the fused kernel's pi_loss)
    loss_tie_rule = 0       # ARL_PPO_TIE_THEANO; only PPO's surrogate has a tie to break (BasePPO.ppo_tie_rule)

    def pi_loss(self, policy, act, adv, old_dist_info, new_dist_info, valids):
        """The subclass's policy loss as a formula on tensors (a2c.py:43-46 / ppo.py:42-51).  The learner does
        not evaluate it -- `loss_kind` picks the same expression inside the fused head kernel; the numerics
        tests differentiate it with autograd as the kernel's reference (tests/autograd_ref.py)."""
        raise NotImplementedError

    @property
    def opt_info_keys(self):
        return ["GradNorm"]
