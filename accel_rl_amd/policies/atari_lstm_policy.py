"""Recurrent Atari policies: conv stack -> one recurrent layer -> (softmax pi, linear V)   (SURVEY 8 f3).

`RecurrentCnnPolicy` is the part the reference's AtariLstmPolicy / AtariGruPolicy / AtariRnnPolicy
share (accel_rl/policies/pg/atari_{lstm,gru,rnn}_policy.py, policies/base.py:32-93): the hidden
state kept per environment across steps, reset per environment when the sampler reports a reset,
`agent_infos` carrying the PREVIOUS state of every step, training by BPTT over each environment's
segment of the batch from its stored initial state (only `s[::horizon]` is used,
aac_base.py:157-161).  A cell supplies its parameter layout and the per-step arithmetic.

`AtariLstmPolicy` mirrors PgCnnLstm / FastLstmLayer (pg/networks/pg_cnn_lstm.py:10-165,
policies/layers.py:292-385): gate order f, i, c~, o, sigmoid gates, tanh cell / output
nonlinearity, h0 = c0 = 0 (not trainable).

The matrix products are the dense fp32-MFMA kernels (x W_x + b for all steps at once, h W_h per
step); csrc/lstm.hip and csrc/gru.hip do the gate arithmetic and its backward in place on time
slices of [trajectory][time] arrays; everything else is AtariCnnPolicy's.  The hidden state lives
on the device (the reference keeps it on the CPU and re-uploads it every step).  Stacked
recurrent layers (pg_cnn_rnn_double.py) are not built.
"""
import numpy as np
import torch

from accel_rl_amd import _lib
from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy, _norm_c


class RecurrentCnnPolicy(AtariCnnPolicy):
    """Cell hooks: `_gate_mult` (width of gx / gh in units of H), `_saved_mult` (per-step values kept
    for the backward pass), `_state_keys`, the four `_hidden_*` parameter-layout hooks (which must
    set self._k_x = index of the internal [W_x^T, W_h^T, b] triple relative to the first hidden
    tensor), `_cell_fwd` and `_cell_bwd`."""

    serves_rows = False         # own prob_value: the sampler keeps a contiguous copy of the current observations
    _gate_mult = 4
    _saved_mult = 4
    _separate_dgh = False       # gradient wrt h_prev W_h differs from the one wrt x W_x (GRU)
    _state_keys = ("hprev_0", "cprev_0")
    _k_rel = 0                  # internal index of W_x^T among the hidden tensors

    def __init__(self, conv_filters, conv_filter_sizes, conv_strides, conv_pads, hidden_sizes=(), pixel_scale=255.,
                 alternating_sampler=False, initial_param_values=None):
        if len(hidden_sizes) != 1:
            raise NotImplementedError("exactly one recurrent layer is supported (got hidden_sizes=%r)" % (hidden_sizes,))
        super().__init__(conv_filters, conv_filter_sizes, conv_strides, conv_pads, hidden_sizes=hidden_sizes,
                         pixel_scale=pixel_scale, initial_param_values=initial_param_values)
        self._H = int(hidden_sizes[0])
        if self._H % 4 or self._H > 1024:
            raise NotImplementedError("recurrent width must be a multiple of 4 and <= 1024")

    recurrent = property(lambda self: True)
    state_info_keys = property(lambda self: list(self._state_keys))

    def initialize(self, env_spec, device=None, alternating_sampler=False, **kwargs):
        super().initialize(env_spec, device=device, **kwargs)
        self._k_x = 2 * self._n_conv + self._k_rel      # indices of W_x^T, W_h^T, b in params / grads
        self._state = None

    # ---- recurrent state (policies/base.py:50-93) ------------------------------
    def reset(self, n_batch=None):
        if n_batch is None:
            return
        self._state = [torch.zeros((n_batch, self._H), dtype=torch.float32, device=self.device)
                       for _ in self._state_keys]

    def reset_one(self, idx):
        for s in self._state:
            s[idx] = 0.

    def reset_rows(self, mask_u8):
        """Device form of reset_one for every env whose flag is set (initial state = 0)."""
        keep = (mask_u8 == 0).to(torch.float32).unsqueeze(1)
        for s in self._state:
            s.mul_(keep)

    def get_prev_hiddens(self):
        return list(self._state)

    get_state_info = get_prev_hiddens

    # ---- forward ---------------------------------------------------------------
    def _geom(self, rows, fan_in, units):
        key = ("rec", rows, fan_in, units, _lib.default_route)      # the route is a field of the geometry
        if key not in self._geoms:
            self._geoms[key] = _lib.dense_geom(rows, fan_in, units)
        return self._geoms[key]

    def _conv_features(self, x, tag=""):
        """conv stack only -> (conv activations, [rows, fan] view of the last one)."""
        b = x.shape[0]
        acts, _ = self._trunk(x, tag=tag)          # no dense hidden layers: _hid_geom is empty
        return acts, acts[-1].view(b, -1)

    def _step(self, observations, prev, tag="s"):
        b, hh, gm = observations.shape[0], self._H, self._gate_mult
        w, k = self._w, self._k_x
        _, xf = self._conv_features(self._scaled(observations, tag=tag), tag=tag)
        gx = self._buffer(("gx" + tag, b), (b, gm * hh))
        gh = self._buffer(("gh" + tag, b), (b, gm * hh))
        _lib.conv2d_fwd(xf, w[k], w[k + 2], gx, self._geom(b, self._rec_fan, gm * hh), False, self._conv_ws)
        _lib.conv2d_fwd(prev[0], w[k + 1], None, gh, self._geom(b, hh, gm * hh), False, self._conv_ws)
        new = [self._buffer(("new%d" % i + tag, b), (b, hh)) for i in range(len(prev))]
        self._cell_fwd(gx, gh, prev, new, None)
        prob = torch.empty((b, self.n_act), dtype=torch.float32, device=self.device)
        value = torch.empty(b, dtype=torch.float32, device=self.device)
        _lib.pg_head_infer(new[0], self.params[self._k_head], self.params[self._k_head + 1], prob, value)
        return prob, value, new

    def prob_value(self, observations, state_infos=None):
        """Does NOT advance the internal state (atari_lstm_policy.py:122-139)."""
        with torch.no_grad():
            prev = self._state if state_infos is None else list(state_infos)
            prob, value, _ = self._step(observations, prev, tag="v")
            return prob, value

    def act_step(self, observations, rows=None):
        """The sampler's serving call: (prob, value, *previous state) and the state advances
        (get_actions, atari_lstm_policy.py:161-169).  The returned previous state is only valid
        until the next act_step.  rows = (lo, hi): the observations are those of state rows lo .. hi - 1 only (a
        sampler serving its envs in groups: the reference's pair of per-group states, policies/base.py:44-93, kept
        side by side in one tensor)."""
        with torch.no_grad():
            b = observations.shape[0]
            state = self._state if rows is None else [s[rows[0]:rows[1]] for s in self._state]
            if state[0].shape[0] != b:
                raise ValueError("act_step: %d observations for %d state rows" % (b, state[0].shape[0]))
            ret = [self._buffer(("prev_ret%d" % i, b), (b, self._H)) for i in range(len(state))]
            for r, s in zip(ret, state):
                r.copy_(s)
            prob, value, new = self._step(observations, state)
            for s, n in zip(state, new):
                s.copy_(n)
            return (prob, value) + tuple(ret)

    def value(self, observations, state_infos=None):
        return self.prob_value(observations, state_infos)[1]

    def get_actions(self, observations):
        prob, value, *prev = self.act_step(observations)
        infos = dict(prob=prob, value=value)
        infos.update({k: p.clone() for k, p in zip(self._state_keys, prev)})
        return self._sample(prob), infos

    def get_action(self, observation, deterministic=False):
        if self._state is None or self._state[0].shape[0] != 1:
            self.reset(n_batch=1)
        prob, value, *prev = self.act_step(observation[None])
        action = torch.argmax(prob[0]) if deterministic else self._sample(prob)[0]
        infos = dict(prob=prob[0], value=value[0])
        infos.update({k: p[0].clone() for k, p in zip(self._state_keys, prev)})
        return action, infos

    # ---- training: BPTT over each environment's segment ----------------------------
    def loss_and_grads(self, mb, kind, clip_param, v_loss_coeff, ent_loss_coeff, lr_mult, inv_count=None,
                       tie_rule=_lib.PPO_TIE_THEANO):
        """Whole-batch update (rows env-major = [trajectory][time]); mb additionally carries
        `horizon` and the stored previous states (only the rows of t = 0 are used)."""
        if mb.get("idx") is not None:
            raise NotImplementedError("recurrent training takes whole trajectories: no row minibatches")
        with torch.no_grad():
            t_len, hh, gm, n_state = int(mb["horizon"]), self._H, self._gate_mult, len(self._state_keys)
            x = self._scaled(mb["observations"])
            rows = x.shape[0]
            nb = rows // t_len
            assert nb * t_len == rows
            w, g, k, kh = self._w, self.grads, self._k_x, self._k_head
            acts, xf = self._conv_features(x)
            buf = lambda name, cols: self._buffer((name, rows), (rows, cols))            # noqa: E731
            gx, dgx = buf("gx_all", gm * hh), buf("dgx_all", gm * hh)
            dgh = buf("dgh_all", gm * hh) if self._separate_dgh else dgx
            saved = buf("saved_all", self._saved_mult * hh) if self._saved_mult else None
            st_all = [buf("state%d_all" % i, hh) for i in range(n_state)]
            hprev_all = buf("hprev_all", hh)
            g_xh = self._geom(rows, self._rec_fan, gm * hh)
            _lib.conv2d_fwd(xf, w[k], w[k + 2], gx, g_xh, False, self._conv_ws)
            sl = lambda a, t: a.view(nb, t_len, -1)[:, t]                                 # noqa: E731  time slice, strided rows
            init = [sl(mb[key], 0) for key in self._state_keys]
            prev_at = lambda t: init if t == 0 else [sl(s, t - 1) for s in st_all]        # noqa: E731
            hp = self._buffer(("hp", nb), (nb, hh))
            gh = self._buffer(("gh", nb), (nb, gm * hh))
            g_hh = self._geom(nb, hh, gm * hh)
            for t in range(t_len):                                                        # forward scan
                prev = prev_at(t)
                hp.copy_(prev[0])
                sl(hprev_all, t).copy_(hp)
                _lib.conv2d_fwd(hp, w[k + 1], None, gh, g_hh, False, self._conv_ws)
                self._cell_fwd(sl(gx, t), gh, prev, [sl(s, t) for s in st_all], None if saved is None else sl(saved, t))
            # ---- heads + losses on every step
            dout = self._buffer(("dout", rows), (rows, self.n_act + 1))
            dh_all = buf("dh_all", hh)
            loss4 = self._buffer(("loss", rows), (4,))
            _lib.pg_head_loss(st_all[0], self.params[kh], self.params[kh + 1], mb["actions"], mb["advantages"],
                              mb["returns"], mb.get("old_prob"), mb.get("valids"), None, lr_mult, inv_count,
                              self.n_act, kind, clip_param, v_loss_coeff, ent_loss_coeff, dout, dh_all, g[kh],
                              g[kh + 1], loss4, self._loss_ws, tie_rule=tie_rule)
            # ---- backward scan
            carry = self._buffer(("carry", nb), (nb, hh))
            dh_rec = self._buffer(("dh_rec", nb), (nb, hh))
            dg_t = self._buffer(("dg_t", nb), (nb, gm * hh))
            for t in range(t_len - 1, -1, -1):
                last = t == t_len - 1
                self._cell_bwd(sl(dh_all, t), None if last else dh_rec, carry, last,
                               None if saved is None else sl(saved, t), prev_at(t), [sl(s, t) for s in st_all],
                               sl(dgx, t), sl(dgh, t))
                if t > 0:
                    dg_t.copy_(sl(dgh, t))
                    _lib.conv2d_bwd_data(dg_t, w[k + 1], None, dh_rec, g_hh)
            # ---- parameter gradients of the recurrent layer, then the conv stack
            _lib.conv2d_bwd_weight(dgh, hprev_all, self._g[k + 1], self._geom(rows, hh, gm * hh), self._conv_ws)
            _lib.conv2d_bwd_weight(dgx, xf, self._g[k], g_xh, self._conv_ws)
            dxf = buf("dxf", self._rec_fan)
            _lib.conv2d_bwd_data(dgx, w[k], None, dxf, g_xh)
            # db = column sums of dgx = the weight gradient of a 4-channel all-ones input (column 0)
            ones = buf("ones4", 4)
            if not getattr(ones, "_filled", False):
                ones.fill_(1.)
                ones._filled = True
            db4 = self._buffer(("db4", gm * hh), (gm * hh, 4))
            _lib.conv2d_bwd_weight(dgx, ones, db4, self._geom(rows, 4, gm * hh), self._conv_ws)
            g[k + 2].copy_(db4[:, 0])
            self._backward_convs(x, acts, dxf.view(acts[-1].shape))
            return loss4


class AtariLstmPolicy(RecurrentCnnPolicy):

    _gate_mult, _saved_mult, _separate_dgh = 4, 4, False
    _state_keys = ("hprev_0", "cprev_0")

    # ---- parameters: W_x (fan, 4H), W_h (H, 4H), b (4H) in the reference's order (layers.py:325-327)
    def _hidden_reference_init(self, fan):
        h = self._H
        self._hid_geom, self._rec_fan = [], fan
        return [_norm_c((fan, 4 * h), 1.0), _norm_c((h, 4 * h), 1.0), np.zeros(4 * h, np.float32)], \
               ["LstmWx", "LstmWh", "Lstmb"], h

    def _hidden_internal_shapes(self):
        h = self._H
        return [(4 * h, self._rec_fan), (4 * h, h), (4 * h,)]

    def _hidden_to_reference(self, arrs):
        return [self._conv_flat_to_reference(arrs[0]), arrs[1].T, arrs[2]]

    def _hidden_to_internal(self, refs):
        return [self._conv_flat_to_internal(refs[0]), refs[1].T, refs[2]]

    def _cell_fwd(self, gx, gh, prev, out, saved):
        _lib.lstm_cell_fwd(gx, gh, prev[1], out[0], out[1], saved)

    def _cell_bwd(self, dh, dh_rec, carry, last, saved, prev, out, dgx, dgh):
        # carry = dc of the next step (in), dc of this step (out)
        _lib.lstm_cell_bwd(dh, dh_rec, None if last else carry, saved, prev[1], out[1], dgx, carry)
