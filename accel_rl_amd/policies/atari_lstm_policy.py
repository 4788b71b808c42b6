"""Recurrent Atari policy: conv stack -> LSTM -> (softmax pi, linear V)   (SURVEY 8 f3).

Mirror of the reference's AtariLstmPolicy / PgCnnLstm / FastLstmLayer
(accel_rl/policies/pg/atari_lstm_policy.py:15-180, pg/networks/pg_cnn_lstm.py:10-165,
policies/layers.py:292-385, policies/base.py:32-93): one LSTM layer directly on the conv
output, gate order f, i, c~, o, sigmoid gates, tanh cell / output nonlinearity, h0 = c0 = 0
(not trainable), hidden state kept per environment across steps, reset per environment when the
sampler reports a reset, `agent_infos` carrying the PREVIOUS (h, c) of every step, training by
BPTT over each environment's segment of the batch from its stored initial state (only
`s[::horizon]` is used, aac_base.py:157-161).

The matrix products are the dense fp32-MFMA kernels (x W_x + b for all steps at once, h W_h per
step); csrc/lstm.hip does the gate arithmetic and its backward in place on time slices of
[trajectory][time] arrays; everything else is AtariCnnPolicy's.  The hidden state lives on the
device (the reference keeps it on the CPU and re-uploads it every step).  GRU / vanilla-RNN
variants and stacked recurrent layers are not built.
"""
import numpy as np
import torch

from accel_rl_amd import _lib
from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy, _norm_c


class AtariLstmPolicy(AtariCnnPolicy):

    def __init__(self, conv_filters, conv_filter_sizes, conv_strides, conv_pads, hidden_sizes=(), pixel_scale=255.,
                 alternating_sampler=False, initial_param_values=None):
        if len(hidden_sizes) != 1:
            raise NotImplementedError("exactly one recurrent layer is supported (got hidden_sizes=%r)" % (hidden_sizes,))
        super().__init__(conv_filters, conv_filter_sizes, conv_strides, conv_pads, hidden_sizes=hidden_sizes,
                         pixel_scale=pixel_scale, initial_param_values=initial_param_values)
        self._H = int(hidden_sizes[0])
        if self._H % 4 or self._H > 1024:
            raise NotImplementedError("LSTM width must be a multiple of 4 and <= 1024")

    recurrent = property(lambda self: True)
    state_info_keys = property(lambda self: ["hprev_0", "cprev_0"])

    # ---- parameters: W_x (fan, 4H), W_h (H, 4H), b (4H) in the reference's order (layers.py:325-327)
    def _hidden_reference_init(self, fan):
        h = self._H
        self._hid_geom, self._lstm_fan = [], fan
        return [_norm_c((fan, 4 * h), 1.0), _norm_c((h, 4 * h), 1.0), np.zeros(4 * h, np.float32)], \
               ["LstmWx", "LstmWh", "Lstmb"], h

    def _hidden_internal_shapes(self):
        h = self._H
        return [(4 * h, self._lstm_fan), (4 * h, h), (4 * h,)]

    def _hidden_to_reference(self, arrs):
        return [self._conv_flat_to_reference(arrs[0]), arrs[1].T, arrs[2]]

    def _hidden_to_internal(self, refs):
        return [self._conv_flat_to_internal(refs[0]), refs[1].T, refs[2]]

    def initialize(self, env_spec, device=None, alternating_sampler=False, **kwargs):
        super().initialize(env_spec, device=device, **kwargs)
        self._k_x = 2 * self._n_conv            # indices of W_x, W_h, b in params / grads
        self._h = self._c = None

    # ---- recurrent state (policies/base.py:50-93) ------------------------------
    def reset(self, n_batch=None):
        if n_batch is None:
            return
        self._h = torch.zeros((n_batch, self._H), dtype=torch.float32, device=self.device)
        self._c = torch.zeros_like(self._h)

    def reset_one(self, idx):
        self._h[idx] = 0.
        self._c[idx] = 0.

    def reset_rows(self, mask_u8):
        """Device form of reset_one for every env whose flag is set (h0 = c0 = 0)."""
        keep = (mask_u8 == 0).to(torch.float32).unsqueeze(1)
        self._h.mul_(keep)
        self._c.mul_(keep)

    def get_prev_hiddens(self):
        return [self._h, self._c]

    get_state_info = get_prev_hiddens

    # ---- forward ---------------------------------------------------------------
    def _geom(self, rows, fan_in, units):
        key = ("lstm", rows, fan_in, units)
        if key not in self._geoms:
            self._geoms[key] = _lib.dense_geom(rows, fan_in, units)
        return self._geoms[key]

    def _conv_features(self, x, tag=""):
        """conv stack only -> (conv activations, [rows, fan] view of the last one)."""
        b = x.shape[0]
        acts, _ = self._trunk(x, tag=tag)          # no dense hidden layers: _hid_geom is empty
        return acts, acts[-1].view(b, -1)

    def _step(self, observations, h_prev, c_prev, tag="s"):
        b, hh = observations.shape[0], self._H
        w, k = self._w, self._k_x
        _, xf = self._conv_features(self._scaled(observations, tag=tag), tag=tag)
        gx = self._buffer(("gx" + tag, b), (b, 4 * hh))
        gh = self._buffer(("gh" + tag, b), (b, 4 * hh))
        _lib.conv2d_fwd(xf, w[k], w[k + 2], gx, self._geom(b, self._lstm_fan, 4 * hh), False, self._conv_ws)
        _lib.conv2d_fwd(h_prev, w[k + 1], None, gh, self._geom(b, hh, 4 * hh), False, self._conv_ws)
        h_new = self._buffer(("h_new" + tag, b), (b, hh))
        c_new = self._buffer(("c_new" + tag, b), (b, hh))
        _lib.lstm_cell_fwd(gx, gh, c_prev, h_new, c_new)
        prob = torch.empty((b, self.n_act), dtype=torch.float32, device=self.device)
        value = torch.empty(b, dtype=torch.float32, device=self.device)
        _lib.pg_head_infer(h_new, self.params[self._k_head], self.params[self._k_head + 1], prob, value)
        return prob, value, h_new, c_new

    def prob_value(self, observations, state_infos=None):
        """Does NOT advance the internal state (atari_lstm_policy.py:122-139)."""
        with torch.no_grad():
            h, c = (self._h, self._c) if state_infos is None else state_infos
            prob, value, _, _ = self._step(observations, h, c, tag="v")
            return prob, value

    def act_step(self, observations):
        """The sampler's serving call: (prob, value, h_prev, c_prev) and the state advances
        (get_actions, atari_lstm_policy.py:161-169).  The returned previous state is only valid
        until the next act_step."""
        with torch.no_grad():
            b = observations.shape[0]
            hp = self._buffer(("hp_ret", b), (b, self._H))
            cp = self._buffer(("cp_ret", b), (b, self._H))
            hp.copy_(self._h)
            cp.copy_(self._c)
            prob, value, h_new, c_new = self._step(observations, self._h, self._c)
            self._h.copy_(h_new)
            self._c.copy_(c_new)
            return prob, value, hp, cp

    def value(self, observations, state_infos=None):
        return self.prob_value(observations, state_infos)[1]

    def get_actions(self, observations):
        prob, value, hp, cp = self.act_step(observations)
        return self._sample(prob), dict(prob=prob, value=value, hprev_0=hp.clone(), cprev_0=cp.clone())

    def get_action(self, observation, deterministic=False):
        if self._h is None or self._h.shape[0] != 1:
            self.reset(n_batch=1)
        prob, value, hp, cp = self.act_step(observation[None])
        action = torch.argmax(prob[0]) if deterministic else self._sample(prob)[0]
        return action, dict(prob=prob[0], value=value[0], hprev_0=hp[0].clone(), cprev_0=cp[0].clone())

    # ---- training: BPTT over each environment's segment ----------------------------
    def loss_and_grads(self, mb, kind, clip_param, v_loss_coeff, ent_loss_coeff, lr_mult, inv_count=None):
        """Whole-batch update (rows env-major = [trajectory][time]); mb additionally carries
        `horizon` and the stored previous states `hprev_0`, `cprev_0` (only the rows of t = 0 are used)."""
        if mb.get("idx") is not None:
            raise NotImplementedError("recurrent training takes whole trajectories: no row minibatches")
        with torch.no_grad():
            t_len, hh = int(mb["horizon"]), self._H
            x = self._scaled(mb["observations"])
            rows = x.shape[0]
            nb = rows // t_len
            assert nb * t_len == rows
            w, g, k, kh = self._w, self.grads, self._k_x, self._k_head
            acts, xf = self._conv_features(x)
            buf = lambda name, cols: self._buffer((name, rows), (rows, cols))            # noqa: E731
            gx, gates, dgates = buf("gx_all", 4 * hh), buf("gates_all", 4 * hh), buf("dgates_all", 4 * hh)
            h_all, c_all, hprev_all = buf("h_all", hh), buf("c_all", hh), buf("hprev_all", hh)
            _lib.conv2d_fwd(xf, w[k], w[k + 2], gx, self._geom(rows, self._lstm_fan, 4 * hh), False, self._conv_ws)
            sl = lambda a, t: a.view(nb, t_len, -1)[:, t]                                 # noqa: E731  time slice, strided rows
            h0, c0 = sl(mb["hprev_0"], 0), sl(mb["cprev_0"], 0)
            hp = self._buffer(("hp", nb), (nb, hh))
            gh = self._buffer(("gh", nb), (nb, 4 * hh))
            g_hh = self._geom(nb, hh, 4 * hh)
            for t in range(t_len):                                                        # forward scan
                hp.copy_(h0 if t == 0 else sl(h_all, t - 1))
                sl(hprev_all, t).copy_(hp)
                _lib.conv2d_fwd(hp, w[k + 1], None, gh, g_hh, False, self._conv_ws)
                _lib.lstm_cell_fwd(sl(gx, t), gh, c0 if t == 0 else sl(c_all, t - 1), sl(h_all, t), sl(c_all, t),
                                   sl(gates, t))
            # ---- heads + losses on every step
            dout = self._buffer(("dout", rows), (rows, self.n_act + 1))
            dh_all = buf("dh_all", hh)
            loss4 = self._buffer(("loss", rows), (4,))
            _lib.pg_head_loss(h_all, self.params[kh], self.params[kh + 1], mb["actions"], mb["advantages"],
                              mb["returns"], mb.get("old_prob"), mb.get("valids"), None, lr_mult, inv_count,
                              self.n_act, kind, clip_param, v_loss_coeff, ent_loss_coeff, dout, dh_all, g[kh],
                              g[kh + 1], loss4, self._loss_ws)
            # ---- backward scan
            dc = self._buffer(("dc", nb), (nb, hh))
            dh_rec = self._buffer(("dh_rec", nb), (nb, hh))
            dg_t = self._buffer(("dg_t", nb), (nb, 4 * hh))
            for t in range(t_len - 1, -1, -1):
                last = t == t_len - 1
                _lib.lstm_cell_bwd(sl(dh_all, t), None if last else dh_rec, None if last else dc, sl(gates, t),
                                   c0 if t == 0 else sl(c_all, t - 1), sl(c_all, t), sl(dgates, t), dc)
                if t > 0:
                    dg_t.copy_(sl(dgates, t))
                    _lib.conv2d_bwd_data(dg_t, w[k + 1], None, dh_rec, g_hh)
            # ---- parameter gradients of the LSTM, then the conv stack
            g_xh = self._geom(rows, self._lstm_fan, 4 * hh)
            _lib.conv2d_bwd_weight(dgates, hprev_all, self._g[k + 1], self._geom(rows, hh, 4 * hh), self._conv_ws)
            _lib.conv2d_bwd_weight(dgates, xf, self._g[k], g_xh, self._conv_ws)
            dxf = buf("dxf", self._lstm_fan)
            _lib.conv2d_bwd_data(dgates, w[k], None, dxf, g_xh)
            # db = column sums of dgates = the weight gradient of a 4-channel all-ones input (column 0)
            ones = buf("ones4", 4)
            if not getattr(ones, "_filled", False):
                ones.fill_(1.)
                ones._filled = True
            db4 = self._buffer(("db4", 4 * hh), (4 * hh, 4))
            _lib.conv2d_bwd_weight(dgates, ones, db4, self._geom(rows, 4, 4 * hh), self._conv_ws)
            g[k + 2].copy_(db4[:, 0])
            self._backward_convs(x, acts, dxf.view(acts[-1].shape))
            return loss4
