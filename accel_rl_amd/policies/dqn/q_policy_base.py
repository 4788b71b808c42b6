"""What the DQN-family policies share on top of AtariCnnPolicy's trunk: one more dense MFMA call as
the output layer (rows padded to the kernels' tile widths, padding = zero weights and zero
gradients), a target network as a second flat bucket (`update_target` is one device copy),
epsilon-greedy action serving and the output layer's backward pass.

Epsilon-greedy draws stay on the host RNG in the reference's order -- per (step, group):
np.random.rand(B), then action_space.sample_n(#random)
(policies/dqn/atari_dqn_policy.py:125-130, atari_cat_dqn_policy.py:118-124) -- but are made for a
whole rollout at once (`host_draws`) and shipped as an override table, so serving an action needs
no host round trip.
"""
import numpy as np
import torch

from accel_rl_amd import _lib
from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy


class QPolicyBase(AtariCnnPolicy):
    """Subclasses give `_head_width` (columns of the output layer as stored), `_serve(out, override, onehot,
    greedy)` (the action kernel) and the four `_head_*` layout hooks."""

    _epsilon = 1

    @property
    def _head_width(self):
        raise NotImplementedError

    def _serve(self, out, override, onehot, greedy=None):
        raise NotImplementedError

    def initialize(self, env_spec, device=None, **kwargs):
        super().initialize(env_spec, device=device, **kwargs)
        self.flat_target = self.flat_params.clone()             # target network (:57-61)
        sizes = [int(np.prod(s)) for s in self._shapes]
        self._w_target = [self.flat_target[o:o + n] for o, n in zip(self._offsets, sizes)]
        self._overrides = dict()          # n_envs -> (pinned host, device) i32[horizon][n_envs]
        self._step = 0


    # ---- forward -----------------------------------------------------------
    def _logits(self, x, w=None, tag=""):
        """[B, head width] output-layer pre-activations (+ the trunk's activations)."""
        w = self._w if w is None else w
        b = x.shape[0]
        acts, hids = self._trunk(x, w=w, tag=tag)
        k = self._k_head
        out = self._buffer(("logits" + tag, b), (b, self._head_width))
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
            self._geoms[key] = _lib.dense_geom(b, self._hid_geom[-1][0], self._head_width)
        return self._geoms[key]

    def prob_value(self, observations):
        """The sampler's serving call: a one-hot 'prob' row for the epsilon-greedy action of this
        step (so that the categorical sampling kernel picks it) and a zero 'value'."""
        with torch.no_grad():
            b = observations.shape[0]
            logits, _, _ = self._logits(self._scaled(observations))
            onehot = torch.empty((b, self.n_act), dtype=torch.float32, device=self.device)
            ov = None
            if b in self._overrides and self._step < self._overrides[b][1].shape[0]:
                ov = self._overrides[b][1][self._step]
            self._serve(logits, ov, onehot)
            if not hasattr(self, "_zero_value") or self._zero_value.numel() != b:
                self._zero_value = torch.zeros(b, dtype=torch.float32, device=self.device)
            return onehot, self._zero_value

    def greedy_actions(self, observations):
        with torch.no_grad():
            b = observations.shape[0]
            logits, _, _ = self._logits(self._scaled(observations))
            onehot = torch.empty((b, self.n_act), dtype=torch.float32, device=self.device)
            greedy = torch.empty(b, dtype=torch.uint8, device=self.device)
            self._serve(logits, None, onehot, greedy)
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

    # ---- training: the output layer's backward, then the trunk's ---------------------
    def _head_backward(self, dout, x, acts, hids):
        """dW = dout^T h, db = column sums of dout (riding along in the weight-gradient kernel),
        dh = (dout W) * (h > 0) -- one launch; the folds run at the end of the trunk's backward."""
        b = x.shape[0]
        k = self._k_head
        geom = self._head_geom(b)
        hid = self._hid_geom[-1][0]
        dh = self._buffer(("dh", b), (b, hid))
        done = self._folds.conv2d_bwd_pair(dout, self._w[k], hids[-1], dh, hids[-1], self._g[k], geom,
                                           self._fold_ws(("dw", k)), dbias=self.grads[k + 1])
        if not done:        # ragged batch (generic kernels): column sums as the weight gradient of an all-ones input
            ones = self._buffer(("ones4", b), (b, 4))
            ones.fill_(1.)
            db4 = self._buffer(("db4", b), (dout.shape[1], 4))
            _lib.conv2d_bwd_weight(dout, ones, db4, self._ones_geom(b, dout.shape[1]), self._conv_ws)
            self.grads[k + 1].copy_(db4[:, 0])
        self._backward_trunk(x, acts, hids, dh, masked=True)
