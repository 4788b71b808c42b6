"""What the DQN-family policies share on top of AtariCnnPolicy's trunk: one more dense MFMA call as
the output layer (rows padded to the kernels' tile widths, padding = zero weights and zero
gradients), a target network as a second flat bucket (`update_target` is one device copy),
epsilon-greedy action serving and the output layer's backward pass.

Epsilon-greedy draws stay on the host RNG in the reference's order -- per (step, group):
np.random.rand(B), then action_space.sample_n(#random)
(policies/dqn/atari_dqn_policy.py:125-130, atari_cat_dqn_policy.py:118-124) -- but are made for a
whole rollout at once (`host_draws`) and shipped as an override table, so serving an action needs
no host round trip.

Dueling networks (dqn_cnn.py:89-112, catdqn_cnn.py:77-93; one hidden layer): the advantage stream's and
the value stream's hidden layers are stacked into ONE dense layer of 2H units (rows 0..H-1 advantage,
H..2H-1 value) and their output layers into ONE block-structured matrix over those 2H inputs
(advantage rows read columns 0..H-1, value rows columns H..2H-1; the off-blocks are exactly zero
and their gradients are masked to zero), so trunk and output layer stay one MFMA call each.  The
merge val + (adv - mean adv) lives in the action / loss kernels.  The flat parameter vector keeps
the reference's order (value branch first: hidden_Val, Val, hidden, output).
"""
import numpy as np
import torch

from accel_rl_amd import _lib
from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy, ObsRows, _norm_c


class QPolicyBase(AtariCnnPolicy):
    """Subclasses give `_head_width` (columns of the output layer as stored), `_serve(out, override, onehot,
    greedy)` (the action kernel) and the four `_head_*` layout hooks."""

    serves_rows = False         # own prob_value: the sampler keeps a contiguous copy of the current observations
    _epsilon = 1
    _dueling = False

    def _set_dueling(self, dueling):
        self._dueling = bool(dueling)
        if self._dueling:
            if len(self.hidden_sizes) != 1:
                raise NotImplementedError("dueling networks are built for one hidden layer (INTEGRATION.md, section E)")
            # construction order [hid W, b, hid_Val W, b, out W, b, Val W, b] -> the reference's flat order
            self._tail_perm = [2, 3, 6, 7, 0, 1, 4, 5]

    # units of the advantage / value output layers (reference shapes)
    def _out_units(self):
        raise NotImplementedError

    def _val_units(self):
        raise NotImplementedError

    def _hidden_reference_init(self, fan):
        if not self._dueling:
            return super()._hidden_reference_init(fan)
        hs = self.hidden_sizes[0]
        if hs % 4:
            raise NotImplementedError("hidden sizes must be multiples of 4 (got %d)" % hs)
        # draws in the reference's construction order: hidden, output, hidden_Val, Val
        hid = [_norm_c((fan, hs), 1.0), np.zeros(hs, np.float32)]
        out = [_norm_c((hs, self._out_units()), 0.01), np.zeros(self._out_units(), np.float32)]
        hid_val = [_norm_c((fan, hs), 1.0), np.zeros(hs, np.float32)]
        val = [_norm_c((hs, self._val_units()), 0.01), np.zeros(self._val_units(), np.float32)]
        self._duel_head_ref = out + val
        self._hid_geom = [(2 * hs, fan)]
        return hid + hid_val, ["FC0W", "FC0b", "FCVal0W", "FCVal0b"], 2 * hs

    def _hidden_to_reference(self, arrs):
        if not self._dueling:
            return super()._hidden_to_reference(arrs)
        hs = self.hidden_sizes[0]
        w, b = arrs
        return [self._conv_flat_to_reference(w[:hs]), b[:hs], self._conv_flat_to_reference(w[hs:]), b[hs:]]

    def _hidden_to_internal(self, refs):
        if not self._dueling:
            return super()._hidden_to_internal(refs)
        return [np.concatenate([self._conv_flat_to_internal(refs[0]), self._conv_flat_to_internal(refs[2])], axis=0),
                np.concatenate([refs[1], refs[3]])]

    def _duel_blocks(self):
        """(rows of the advantage block, rows of the value block) of the stored output matrix."""
        raise NotImplementedError

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
        if self._dueling:                 # 1 on the two blocks of the output matrix that exist, 0 elsewhere
            hs = self.hidden_sizes[0]
            adv_rows, val_rows = self._duel_blocks()
            mask = torch.zeros(self._shapes[self._k_head], dtype=torch.float32, device=self.device)
            mask[adv_rows, :hs] = 1.
            mask[val_rows, hs:] = 1.
            self._duel_mask = mask


    # ---- forward -----------------------------------------------------------
    def _logits(self, x, w=None, tag="", parts_ws=None):
        """[B, head width] output-layer pre-activations (+ the trunk's activations).
        parts_ws (a conv workspace of its own): the output layer's split reduction stays unfolded there and the first
        return value is the ArlFoldItem describing it (for a loss kernel that folds while it reads: _lib.logit_src)."""
        w = self._w if w is None else w
        b = x.shape[0]
        acts, hids = self._trunk(x, w=w, tag=tag)
        k = self._k_head
        out = self._buffer(("logits" + tag, b), (b, self._head_width))
        geom = self._head_geom(b)
        if parts_ws is not None:
            return _lib.conv2d_fwd_parts(hids[-1], w[k], w[k + 1], out, geom, False, parts_ws), acts, hids
        _lib.conv2d_fwd(hids[-1], w[k], w[k + 1], out, geom, False, self._conv_ws)
        return out, acts, hids

    def _head_parts_ws(self):
        """Two workspaces for the output layers' unfolded partial sums (online pass, target pass): they must outlive the
        other launches of the forward passes, which split into the policy's common workspace."""
        if getattr(self, "_head_ws", None) is None:
            self._head_ws = (_lib.conv_workspace(self.device), _lib.conv_workspace(self.device))
        return self._head_ws

    def _pair_rows(self, obs, next_obs):
        """u8 [2B,C,H,W] = obs followed by next_obs: in place when the replay memory handed them out adjacent
        (FrameReplayBuffer._batch_outputs), else through a scratch copy."""
        b = obs.shape[0]
        if (obs.is_contiguous() and next_obs.is_contiguous() and
                obs.untyped_storage().data_ptr() == next_obs.untyped_storage().data_ptr() and
                next_obs.data_ptr() == obs.data_ptr() + obs.numel()):
            return torch.as_strided(obs, (2 * b,) + tuple(obs.shape[1:]), obs.stride())
        key = ("obs_pair", b)
        both = None if torch.cuda.is_current_stream_capturing() else self._scratch.get(key)
        if both is None:
            both = torch.empty((2 * b,) + tuple(obs.shape[1:]), dtype=torch.uint8, device=self.device)
            if not torch.cuda.is_current_stream_capturing():
                self._scratch[key] = both
        both[:b].copy_(obs)
        both[b:].copy_(next_obs)
        return both

    def _forward_for_loss(self, obs, next_obs, double_dqn, head_parts=False):
        """The three forward passes of a DQN-family loss: online net on obs (activations kept for the backward
        pass), target net on next_obs and -- double DQN -- online net on next_obs.  The two online passes run as
        ONE pass over 2B rows (at the DQN batch of 32 every layer is latency-bound, so the second half is nearly
        free), whose first-half slices feed the backward pass; the target pass reads the same scaled next_obs.
        head_parts (taken on the double-DQN path from u8 rows, the one the benchmarks run; ignored elsewhere): the three
        logit entries come back as _lib.ArlLogitSrc -- the output layers' split partial sums, unfolded.
        Returns (x, out, acts, hids, target_out, online_next_out or None)."""
        b = obs.shape[0]
        c, h, w = self._obs_shape
        if double_dqn and self._u8:                     # conv 1 reads the u8 rows itself: no scaled copy at all
            both = self._pair_rows(obs, next_obs)
            if head_parts:          # the two output layers' folds are left to the loss kernel (arl_catdqn_loss_parts)
                ws2, wst = self._head_parts_ws()
                k, r = self._k_head, self._head_width
                it_t, _, _ = self._logits(ObsRows(both[b:], None), w=self._w_target, tag="t", parts_ws=wst)
                it_2, acts2, hids2 = self._logits(ObsRows(both, None), tag="2", parts_ws=ws2)
                return (ObsRows(both[:b], None), _lib.logit_src(it_2, self._w[k + 1], 0, r), [a[:b] for a in acts2],
                        [hd[:b] for hd in hids2], _lib.logit_src(it_t, self._w_target[k + 1], 0, r),
                        _lib.logit_src(it_2, self._w[k + 1], b, r))
            tgt, _, _ = self._logits(ObsRows(both[b:], None), w=self._w_target, tag="t")
            out2, acts2, hids2 = self._logits(ObsRows(both, None), tag="2")
            return (ObsRows(both[:b], None), out2[:b], [a[:b] for a in acts2], [hd[:b] for hd in hids2], tgt,
                    out2[b:])
        if not double_dqn or c != 4:
            tgt, _, _ = self._logits(self._scaled(next_obs, tag="n"), w=self._w_target, tag="t")
            pol_next = self._logits(self._scaled(next_obs, tag="n"), tag="d")[0] if double_dqn else None
            x = self._scaled(obs)
            out, acts, hids = self._logits(x)
            return x, out, acts, hids, tgt, pol_next
        x2 = self._buffer(("x2", 2 * b), (2 * b, c, h, w), channels_last=True)
        _lib.gather_scale_obs_nhwc(obs, None, x2[:b], self._scale)
        _lib.gather_scale_obs_nhwc(next_obs, None, x2[b:], self._scale)
        tgt, _, _ = self._logits(x2[b:], w=self._w_target, tag="t")
        out2, acts2, hids2 = self._logits(x2, tag="2")
        return x2[:b], out2[:b], [a[:b] for a in acts2], [hd[:b] for hd in hids2], tgt, out2[b:]

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

    def serve_group(self, observations, row0, n_envs):
        """prob_value for the B observations of envs [row0, row0 + B) of an n_envs-wide rollout: a sampler that serves
        its envs in groups (HostEnvSampler, the reference's two alternating halves) reads the group's slice of the
        step's row of the override table that host_draws(horizon, n_envs) filled."""
        with torch.no_grad():
            b = observations.shape[0]
            logits, _, _ = self._logits(self._scaled(observations))
            onehot = torch.empty((b, self.n_act), dtype=torch.float32, device=self.device)
            table = self._overrides[n_envs][1]
            if self._step >= table.shape[0] or row0 + b > table.shape[1]:
                raise IndexError("serve_group: step %d / rows %d..%d outside the %s override table" %
                                 (self._step, row0, row0 + b, tuple(table.shape)))
            self._serve(logits, table[self._step, row0:row0 + b], onehot)
            return onehot, torch.zeros(b, dtype=torch.float32, device=self.device)

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
        if self._dueling:                 # the folds have run: keep the absent blocks' gradient at exactly zero
            self.grads[k].mul_(self._duel_mask)
