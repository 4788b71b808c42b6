"""Plain DQN policy for Atari: conv stack -> dense -> one Q value per action; a target network;
epsilon-greedy action serving.

Mirror of the reference's AtariDqnPolicy / DqnCnn
(accel_rl/policies/dqn/atari_dqn_policy.py:15-137, policies/dqn/networks/dqn_cnn.py:11-117).  Trunk,
target network and action serving are QPolicyBase's; the output layer "output_q" is one dense MFMA
call whose row is padded to 32 columns (zero weights, zero gradients) so that its data gradient
runs on the scalar-addressed kernels, followed by csrc/dqn.hip (arl_dqn_act, arl_dqn_loss).
Dueling (`dueling=True`): see QPolicyBase -- the stored row is n_actions advantages followed by the value.
`shared_last_bias=True` (dqn_cnn.py:67-82: the output layer has no bias of its own, a BiasLayer with shared_axes=(0, 1) adds ONE
scalar to every action's value): the stored bias vector keeps one entry per action, all equal -- their gradient is the sum
over the actions, written back to every entry after the backward pass's folds, so an elementwise optimiser keeps them
equal; the reference-layout vector (get / set_param_values) carries the one scalar.
"""
import numpy as np
import torch

from accel_rl_amd import _lib
from accel_rl_amd.policies.atari_cnn_policy import _norm_c
from accel_rl_amd.policies.dqn.q_policy_base import QPolicyBase


class AtariDqnPolicy(QPolicyBase):

    def __init__(self, conv_filters, conv_filter_sizes, conv_strides, conv_pads, hidden_sizes=(),
                 pixel_scale=255., epsilon=1, dueling=False, shared_last_bias=False, initial_param_values=None):
        super().__init__(conv_filters, conv_filter_sizes, conv_strides, conv_pads, hidden_sizes=hidden_sizes,
                         pixel_scale=pixel_scale, initial_param_values=initial_param_values)
        self._epsilon = epsilon
        self._shared_last_bias = bool(shared_last_bias)
        self._set_dueling(dueling)

    # ---- output layer: "output_q" dense, n_actions units (dqn_cnn.py:73-80) [+ "Val", 1 unit (:100-107)]
    def _out_units(self):
        return self.n_act

    def _val_units(self):
        return 1

    def _duel_blocks(self):
        return slice(0, self.n_act), slice(self.n_act, self.n_act + 1)

    def _head_reference_init(self, fan, n_act):
        self._q_stride = (n_act + int(self._dueling) + 31) // 32 * 32
        out_b = np.zeros(1 if self._shared_last_bias else n_act, np.float32)     # (a shared bias is ONE parameter)
        if self._dueling:
            ref = list(self._duel_head_ref)
            ref[1] = out_b
            return ref, ["OutputW", "Outputb", "ValW", "Valb"]
        return [_norm_c((fan, n_act), 0.01), out_b], ["OutputW", "Outputb"]

    def _head_internal_shapes(self, fan, n_act):
        return [(self._q_stride, fan), (self._q_stride,)]

    def _head_to_reference(self, wh, bh):
        a = self.n_act
        out_b = bh[:1] if self._shared_last_bias else bh[:a]        # (gradients / optimiser slots: every entry holds the sum)
        if self._dueling:
            hs = self.hidden_sizes[0]
            return [wh[:a, :hs].T, out_b, wh[a:a + 1, hs:].T, bh[a:a + 1]]
        return [wh[:a].T, out_b]

    def _head_to_internal(self, ref_tail):
        a = self.n_act
        fan = ref_tail[0].shape[0] * (2 if self._dueling else 1)
        w = np.zeros((self._q_stride, fan), np.float32)
        b = np.zeros(self._q_stride, np.float32)
        b[:a] = ref_tail[1]                    # (a shared bias broadcasts its one value to every action's entry)
        if self._dueling:
            hs = self.hidden_sizes[0]
            w[:a, :hs] = ref_tail[0].T
            w[a, hs:] = ref_tail[2][:, 0]
            b[a] = ref_tail[3][0]
        else:
            w[:a] = ref_tail[0].T
        return [w, b]

    @property
    def _head_width(self):
        return self._q_stride

    def _serve(self, out, override, onehot, greedy=None):
        _lib.dqn_act(out, override, self.n_act, onehot, greedy, dueling=self._dueling)

    # ---- host-interface twins of q / target_q (:108-112) ------------------------
    def _merged(self, out):
        adv = out[:, :self.n_act]
        if not self._dueling:
            return adv.clone()
        return out[:, self.n_act:self.n_act + 1] + (adv - adv.mean(dim=1, keepdim=True))

    def q(self, observations):
        with torch.no_grad():
            return self._merged(self._logits(self._scaled(observations))[0])

    def target_q(self, observations):
        with torch.no_grad():
            return self._merged(self._logits(self._scaled(observations), w=self._w_target, tag="t")[0])

    # ---- training ------------------------------------------------------------
    def q_loss_and_grads(self, obs, next_obs, actions, returns, terminals, is_weights, gamma_n, delta_clip,
                         double_dqn=False):
        """One minibatch of DQN.build_loss (dqn.py:137-172): forward of the policy net on obs, of the target
        net (and, for double DQN, the policy net) on next_obs, the (Huber) TD loss, and the full backward
        pass into flat_grads.  Returns (loss_rows f32[B] whose sum is the loss, td_abs f32[B])."""
        with torch.no_grad():
            b = obs.shape[0]
            x, q, acts, hids, tgt_q, pol_next = self._forward_for_loss(obs, next_obs, double_dqn)
            dq = self._buffer(("dlogits", b), tuple(q.shape))
            pack = self._buffer(("loss_td", b), (2, b))         # one buffer: DqnOptimizer's statistics ring takes both rows at once
            loss_rows, td_abs = pack[0], pack[1]
            _lib.dqn_loss(q, tgt_q, pol_next, actions, returns, terminals, is_weights, self.n_act, gamma_n,
                          delta_clip, dq, loss_rows, td_abs, dueling=self._dueling)
            self._head_backward(dq, x, acts, hids)
            if self._shared_last_bias:          # the folds have run: d loss / d (shared scalar) = the sum over the actions
                gb = self.grads[self._k_head + 1]
                gb[:self.n_act] = gb[:self.n_act].sum()
            return loss_rows, td_abs
