"""Categorical ("C51") DQN policy for Atari: conv stack -> dense -> n_actions x n_atoms logits,
softmax over atoms per action; a target network; epsilon-greedy action serving.

Mirror of the reference's AtariCatDqnPolicy / CatDqnCnn
(accel_rl/policies/dqn/atari_cat_dqn_policy.py:15-132, policies/dqn/networks/catdqn_cnn.py:11-128).
The trunk (gather, convs, dense, their gradients) is AtariCnnPolicy's -- hand-written HIP, fp32
MFMA; the output layer is one more dense MFMA call whose rows are padded to a multiple of 4
atoms, followed by csrc/dqn.hip (per-action softmax, Q, greedy / epsilon-greedy action; the C51
loss).  The target network is a second flat bucket; `update_target` is one device copy.

Epsilon-greedy draws stay on the host RNG in the reference's order -- per (step, group):
np.random.rand(B), then action_space.sample_n(#random) (atari_cat_dqn_policy.py:118-124) -- but are
made for a whole rollout at once (`host_draws`) and shipped as an override table, so serving an
action needs no host round trip.  Dueling (`dueling=True`): see QPolicyBase -- the stored block is
n_actions advantage rows followed by one value row of atoms.
"""
import os

import numpy as np
import torch

from accel_rl_amd import _lib
from accel_rl_amd.policies.atari_cnn_policy import _norm_c
from accel_rl_amd.policies.dqn.q_policy_base import QPolicyBase


class AtariCatDqnPolicy(QPolicyBase):

    # the loss launch reads the two output layers' split partial sums and folds them itself (arl_catdqn_loss_parts): two
    # launches fewer per update, same bits (tests/test_catdqn_gpu.py); False: the separate fold launches
    loss_folds_heads = os.environ.get("ARL_LOSS_FOLDS_HEADS", "1") != "0"       # (the env switch: same-box A/B)

    def __init__(self, conv_filters, conv_filter_sizes, conv_strides, conv_pads, hidden_sizes=(),
                 pixel_scale=255., epsilon=1, n_atoms=51, dueling=False, initial_param_values=None):
        if not 2 <= n_atoms <= 64:
            raise NotImplementedError("n_atoms must be in [2, 64]")
        super().__init__(conv_filters, conv_filter_sizes, conv_strides, conv_pads, hidden_sizes=hidden_sizes,
                         pixel_scale=pixel_scale, initial_param_values=initial_param_values)
        self.n_atoms = n_atoms
        self._atom_stride = (n_atoms + 3) // 4 * 4
        self._epsilon = epsilon
        self.z = None
        self._set_dueling(dueling)

    def _out_units(self):
        return self.n_act * self.n_atoms

    def _val_units(self):
        return self.n_atoms

    def _duel_blocks(self):
        s = self._atom_stride
        return slice(0, self.n_act * s), slice(self.n_act * s, (self.n_act + 1) * s)

    # ---- output layer: "action_atoms" dense, n_actions * n_atoms units (catdqn_cnn.py:69-76)
    def _head_reference_init(self, fan, n_act):
        # atoms padded per action (zero weights, zero gradients) until the layer's width is a multiple of the
        # MFMA k-tile: its data gradient then runs on the scalar-addressed kernels (18 x 52 = 936 fell back
        # to the generic one: 94 us per update at batch 32)
        while (self._rows * self._atom_stride) % 32 and self._atom_stride < 64:
            self._atom_stride += 4
        if self._dueling:
            return self._duel_head_ref, ["OutputW", "Outputb", "ValW", "Valb"]
        return [_norm_c((fan, n_act * self.n_atoms), 0.01), np.zeros(n_act * self.n_atoms, np.float32)], \
               ["OutputW", "Outputb"]

    @property
    def _rows(self):
        """Rows of atoms per sample in the stored output: the actions (+ the value row when dueling)."""
        return self.n_act + int(self._dueling)

    def _head_internal_shapes(self, fan, n_act):
        return [(self._rows * self._atom_stride, fan), (self._rows * self._atom_stride,)]

    def _head_to_reference(self, wh, bh):
        a, n, s = self.n_act, self.n_atoms, self._atom_stride
        w3, b2 = wh.reshape(self._rows, s, -1), bh.reshape(self._rows, s)
        if self._dueling:
            hs = self.hidden_sizes[0]
            return [w3[:a, :n, :hs].reshape(a * n, hs).T, b2[:a, :n].reshape(-1), w3[a, :n, hs:].T, b2[a, :n]]
        return [w3[:, :n].reshape(a * n, -1).T, b2[:, :n].reshape(-1)]

    def _head_to_internal(self, ref_tail):
        a, n, s = self.n_act, self.n_atoms, self._atom_stride
        hs = ref_tail[0].shape[0]
        w = np.zeros((self._rows, s, hs * (2 if self._dueling else 1)), np.float32)
        b = np.zeros((self._rows, s), np.float32)
        w[:a, :n, :hs] = ref_tail[0].T.reshape(a, n, hs)
        b[:a, :n] = ref_tail[1].reshape(a, n)
        if self._dueling:
            w[a, :n, hs:] = ref_tail[2].T
            b[a, :n] = ref_tail[3]
        return [w.reshape(self._rows * s, -1), b.reshape(-1)]

    def incorporate_z(self, z):
        """Called by the algorithm while initialising (:69-78): the support of the value distribution."""
        z = np.asarray(z, np.float32)
        assert len(z) == self.n_atoms
        self.z = torch.from_numpy(z).to(self.device)

    @property
    def _head_width(self):
        return self._rows * self._atom_stride

    def _serve(self, out, override, onehot, greedy=None):
        assert self.z is not None, "incorporate_z() first (the algorithm does)"
        _lib.catdqn_act(out, self.z, override, self.n_act, self.n_atoms, onehot, greedy, dueling=self._dueling)

    # ---- training ------------------------------------------------------------
    def cat_loss_and_grads(self, obs, next_obs, actions, returns, terminals, is_weights, v_min, v_max, gamma_n,
                           double_dqn=False):
        """One minibatch of CategoricalDQN.build_loss (cat_dqn.py:40-109): forward of the policy net on
        obs, of the target net (and, for double DQN, the policy net) on next_obs, the C51 loss, and the
        full backward pass into flat_grads.  Returns (loss_rows f32[B] whose sum is the loss, kl f32[B])."""
        with torch.no_grad():
            b = obs.shape[0]
            x, logits, acts, hids, tgt_logits, pol_next = self._forward_for_loss(obs, next_obs, double_dqn,
                                                                                 head_parts=self.loss_folds_heads)
            dlogits = self._buffer(("dlogits", b), (b, self._head_width))
            pack = self._buffer(("loss_kl", b), (2, b))         # one buffer: DqnOptimizer's statistics ring takes both rows at once
            loss_rows, kl = pack[0], pack[1]
            if isinstance(logits, _lib.ArlLogitSrc):            # the output layers' partial sums, folded as they are read
                # (the launch also writes the data gradients' weight copies: the backward pass skips its own launch)
                wts = self._dgrad_weight_items(b) if os.environ.get("ARL_WT_IN_LOSS", "1") != "0" else []     # (A/B switch)
                _lib.catdqn_loss_parts(logits, tgt_logits, pol_next, self.z, actions, returns, terminals, is_weights,
                                       self.n_act, self.n_atoms, self._atom_stride, v_min, v_max, gamma_n, dlogits,
                                       loss_rows, kl, dueling=self._dueling, dgrad_weights=wts)
                self._wt_fresh = bool(wts)
            else:
                _lib.catdqn_loss(logits, tgt_logits, pol_next, self.z, actions, returns, terminals, is_weights,
                                 self.n_act, self.n_atoms, v_min, v_max, gamma_n, dlogits, loss_rows, kl,
                                 dueling=self._dueling)
            try:
                self._head_backward(dlogits, x, acts, hids)
            finally:
                self._wt_fresh = False
            return loss_rows, kl
