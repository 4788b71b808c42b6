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
action needs no host round trip.  Dueling heads are not implemented.
"""
import numpy as np
import torch

from accel_rl_amd import _lib
from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy, _norm_c


class AtariCatDqnPolicy(AtariCnnPolicy):

    def __init__(self, conv_filters, conv_filter_sizes, conv_strides, conv_pads, hidden_sizes=(),
                 pixel_scale=255., epsilon=1, n_atoms=51, dueling=False, initial_param_values=None):
        if dueling:
            raise NotImplementedError("dueling C51 heads (catdqn_cnn.py:77-93) are not built")
        if not 2 <= n_atoms <= 64:
            raise NotImplementedError("n_atoms must be in [2, 64]")
        super().__init__(conv_filters, conv_filter_sizes, conv_strides, conv_pads, hidden_sizes=hidden_sizes,
                         pixel_scale=pixel_scale, initial_param_values=initial_param_values)
        self.n_atoms = n_atoms
        self._atom_stride = (n_atoms + 3) // 4 * 4
        self._epsilon = epsilon
        self.z = None

    # ---- output layer: "action_atoms" dense, n_actions * n_atoms units (catdqn_cnn.py:69-76)
    def _head_reference_init(self, fan, n_act):
        # atoms padded per action (zero weights, zero gradients) until the layer's width is a multiple of the
        # MFMA k-tile: its data gradient then runs on the scalar-addressed kernels (18 x 52 = 936 fell back
        # to the generic one: 94 us per update at batch 32)
        while (n_act * self._atom_stride) % 32 and self._atom_stride < 64:
            self._atom_stride += 4
        return [_norm_c((fan, n_act * self.n_atoms), 0.01), np.zeros(n_act * self.n_atoms, np.float32)], \
               ["OutputW", "Outputb"]

    def _head_internal_shapes(self, fan, n_act):
        return [(n_act * self._atom_stride, fan), (n_act * self._atom_stride,)]

    def _head_to_reference(self, wh, bh):
        a, n, s = self.n_act, self.n_atoms, self._atom_stride
        w = wh.reshape(a, s, -1)[:, :n].reshape(a * n, -1)
        return [w.T, bh.reshape(a, s)[:, :n].reshape(-1)]

    def _head_to_internal(self, ref_tail):
        a, n, s = self.n_act, self.n_atoms, self._atom_stride
        w = np.zeros((a, s, ref_tail[0].shape[0]), np.float32)
        w[:, :n] = ref_tail[0].T.reshape(a, n, -1)
        b = np.zeros((a, s), np.float32)
        b[:, :n] = ref_tail[1].reshape(a, n)
        return [w.reshape(a * s, -1), b.reshape(-1)]

    def initialize(self, env_spec, device=None, **kwargs):
        super().initialize(env_spec, device=device, **kwargs)
        self.flat_target = self.flat_params.clone()             # target network (:57-61)
        sizes = [int(np.prod(s)) for s in self._shapes]
        self._w_target = [self.flat_target[o:o + n] for o, n in zip(self._offsets, sizes)]
        self._overrides = dict()          # n_envs -> (pinned host, device) i32[horizon][n_envs]
        self._step = 0

    def incorporate_z(self, z):
        """Called by the algorithm while initialising (:69-78): the support of the value distribution."""
        z = np.asarray(z, np.float32)
        assert len(z) == self.n_atoms
        self.z = torch.from_numpy(z).to(self.device)

    # ---- forward -----------------------------------------------------------
    def _logits(self, x, w=None, tag=""):
        """[B, n_actions * atom_stride] output-layer pre-activations (+ the trunk's activations)."""
        w = self._w if w is None else w
        b = x.shape[0]
        acts, hids = self._trunk(x, w=w, tag=tag)
        k = self._k_head
        out = self._buffer(("logits" + tag, b), (b, self.n_act * self._atom_stride))
        geom = self._head_geom(b)
        _lib.conv2d_fwd(hids[-1], w[k], w[k + 1], out, geom, False, self._conv_ws)
        return out, acts, hids

    def _ones_geom(self, b, width):
        key = ("ones", b, width)
        if key not in self._geoms:
            self._geoms[key] = _lib.dense_geom(b, 4, width)
        return self._geoms[key]

    def _head_geom(self, b):
        key = ("head", b)
        if key not in self._geoms:
            self._geoms[key] = _lib.dense_geom(b, self._hid_geom[-1][0], self.n_act * self._atom_stride)
        return self._geoms[key]

    def prob_value(self, observations):
        """The sampler's serving call: a one-hot 'prob' row for the epsilon-greedy action of this
        step (so that the categorical sampling kernel picks it) and a zero 'value'."""
        assert self.z is not None, "incorporate_z() first (the algorithm does)"
        with torch.no_grad():
            b = observations.shape[0]
            logits, _, _ = self._logits(self._scaled(observations))
            onehot = torch.empty((b, self.n_act), dtype=torch.float32, device=self.device)
            ov = None
            if b in self._overrides and self._step < self._overrides[b][1].shape[0]:
                ov = self._overrides[b][1][self._step]
            _lib.catdqn_act(logits, self.z, ov, self.n_act, self.n_atoms, onehot)
            if not hasattr(self, "_zero_value") or self._zero_value.numel() != b:
                self._zero_value = torch.zeros(b, dtype=torch.float32, device=self.device)
            return onehot, self._zero_value

    def greedy_actions(self, observations):
        with torch.no_grad():
            b = observations.shape[0]
            logits, _, _ = self._logits(self._scaled(observations))
            onehot = torch.empty((b, self.n_act), dtype=torch.float32, device=self.device)
            greedy = torch.empty(b, dtype=torch.uint8, device=self.device)
            _lib.catdqn_act(logits, self.z, None, self.n_act, self.n_atoms, onehot, greedy)
            return greedy

    # ---- epsilon-greedy draws (host RNG, reference order) ---------------------
    def host_draws(self, horizon, n_envs, n_groups=2):
        """All of one rollout's action randomness: for every (step, group) the reference's
        get_actions draws rand(B) and then sample_n(#(rand < epsilon)).  Returns the uniforms the
        sampler feeds its categorical kernel (0.5: with a one-hot row that selects the hot action)."""
        ov = np.full((horizon, n_envs), -1, np.int32)
        per = n_envs // n_groups
        for s in range(horizon):
            for j in range(n_groups):
                u = np.random.rand(per)
                idx = np.where(u < self._epsilon)[0]
                ov[s, j * per + idx] = np.random.randint(low=0, high=self.n_act, size=len(idx), dtype=np.uint8)
        # one table per env count (training / evaluation), allocated once: a captured rollout graph
        # keeps reading the same device buffer
        if n_envs not in self._overrides or self._overrides[n_envs][1].shape[0] != horizon:
            self._overrides[n_envs] = (torch.zeros(ov.shape, dtype=torch.int32).pin_memory(),
                                       torch.zeros(ov.shape, dtype=torch.int32, device=self.device))
        host, dev = self._overrides[n_envs]
        host.copy_(torch.from_numpy(ov))
        dev.copy_(host, non_blocking=True)
        return np.full(horizon * n_envs, 0.5)

    def set_step(self, s):
        self._step = s

    def get_actions(self, observations, deterministic=False):
        """Host-interface twin of the reference's get_actions (one group of one step)."""
        acts = self.greedy_actions(observations).cpu().numpy()
        if not deterministic:
            idx = np.where(np.random.rand(len(acts)) < self._epsilon)[0]
            acts[idx] = np.random.randint(low=0, high=self.n_act, size=len(idx), dtype=np.uint8)
        return acts, dict()

    def get_action(self, observation, deterministic=False):
        if deterministic or (np.random.rand() > self._epsilon):
            action = int(self.greedy_actions(observation[None])[0].item())
        else:
            action = self.action_space.sample()
        return action, dict()

    def get_epsilon(self):
        return self._epsilon

    def set_epsilon(self, value):
        self._epsilon = value

    def update_target(self):
        self.flat_target.copy_(self.flat_params)

    # ---- training ------------------------------------------------------------
    def cat_loss_and_grads(self, obs, next_obs, actions, returns, terminals, is_weights, v_min, v_max, gamma_n,
                           double_dqn=False):
        """One minibatch of CategoricalDQN.build_loss (cat_dqn.py:40-109): forward of the policy net on
        obs, of the target net (and, for double DQN, the policy net) on next_obs, the C51 loss, and the
        full backward pass into flat_grads.  Returns (loss_rows f32[B] whose sum is the loss, kl f32[B])."""
        with torch.no_grad():
            b = obs.shape[0]
            tgt_logits, _, _ = self._logits(self._scaled(next_obs, tag="n"), w=self._w_target, tag="t")
            pol_next = None
            if double_dqn:
                pol_next = self._logits(self._scaled(next_obs, tag="n"), tag="d")[0]
            x = self._scaled(obs)
            logits, acts, hids = self._logits(x)
            dlogits = self._buffer(("dlogits", b), tuple(logits.shape))
            loss_rows = self._buffer(("loss_rows", b), (b,))
            kl = self._buffer(("kl", b), (b,))
            _lib.catdqn_loss(logits, tgt_logits, pol_next, self.z, actions, returns, terminals, is_weights,
                             self.n_act, self.n_atoms, v_min, v_max, gamma_n, dlogits, loss_rows, kl)
            k = self._k_head
            geom = self._head_geom(b)
            hid = self._hid_geom[-1][0]
            # output layer: dW = dlogits^T h, db = column sums of dlogits (riding along in the weight-gradient
            # kernel), dh = (dlogits W) * (h > 0) -- one launch; the folds run at the end of the trunk's backward
            dh = self._buffer(("dh", b), (b, hid))
            done = self._folds.conv2d_bwd_pair(dlogits, self._w[k], hids[-1], dh, hids[-1], self._g[k], geom,
                                               self._fold_ws(("dw", k)), dbias=self.grads[k + 1])
            if not done:        # ragged batch (generic kernels): column sums as the weight gradient of an all-ones input
                ones = self._buffer(("ones4", b), (b, 4))
                ones.fill_(1.)
                db4 = self._buffer(("db4", b), (dlogits.shape[1], 4))
                _lib.conv2d_bwd_weight(dlogits, ones, db4, self._ones_geom(b, dlogits.shape[1]), self._conv_ws)
                self.grads[k + 1].copy_(db4[:, 0])
            self._backward_trunk(x, acts, hids, dh, masked=True)
            return loss_rows, kl
