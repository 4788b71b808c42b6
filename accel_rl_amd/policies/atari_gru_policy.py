"""conv stack -> GRU -> (softmax pi, linear V)   (SURVEY 8 f3).

Mirror of the reference's AtariGruPolicy / PgCnnGru / GruLayer
(accel_rl/policies/pg/atari_gru_policy.py:15-180, pg/networks/pg_cnn_gru.py:10-160,
policies/layers.py:109-192): reset / update gates sigmoid, candidate tanh,
    r = s(x W_xr + h W_hr + b_r);  u = s(x W_xu + h W_hu + b_u)
    c = tanh(x W_xc + r * (h W_hc) + b_c);  h' = (1 - u) h + u c
state key `hprev_0`, h0 = 0 (not trainable), all W drawn with NormCInit(1.0).

The reference layer also registers W_xh, W_hh, b (layers.py:147-149) that its step never reads;
they are trainable parameters with identically zero gradient, so they are kept in the bucket (first
among the hidden tensors, as in the reference's get_params order) to keep the flat parameter
vector, its norm and the optimiser slots the same length and layout.  Internally the three gates'
weights are stacked as W_x^T [3H, fan], W_h^T [3H, H], b [3H] (order r, u, c) so that each product
is one dense MFMA call.
"""
import numpy as np

from accel_rl_amd import _lib
from accel_rl_amd.policies.atari_cnn_policy import _norm_c
from accel_rl_amd.policies.atari_lstm_policy import RecurrentCnnPolicy


class AtariGruPolicy(RecurrentCnnPolicy):

    _gate_mult, _saved_mult, _separate_dgh = 3, 4, True
    _state_keys = ("hprev_0",)
    _k_rel = 3                      # after the three unused tensors

    def _hidden_reference_init(self, fan):
        h = self._H
        self._hid_geom, self._rec_fan = [], fan
        ref, names = [], []
        for gate in ("h", "r", "u", "c"):           # add_param order, layers.py:147-161
            ref += [_norm_c((fan, h), 1.0), _norm_c((h, h), 1.0), np.zeros(h, np.float32)]
            names += ["GruWx" + gate, "GruWh" + gate, "Grub" + gate]
        return ref, names, h

    def _hidden_internal_shapes(self):
        h, fan = self._H, self._rec_fan
        return [(fan, h), (h, h), (h,), (3 * h, fan), (3 * h, h), (3 * h,)]

    def _hidden_to_reference(self, arrs):
        h = self._H
        out = [arrs[0], arrs[1], arrs[2]]
        for j in range(3):
            rows = slice(j * h, (j + 1) * h)
            out += [self._conv_flat_to_reference(arrs[3][rows]), arrs[4][rows].T, arrs[5][rows]]
        return out

    def _hidden_to_internal(self, refs):
        wx = np.concatenate([self._conv_flat_to_internal(refs[3 + 3 * j]) for j in range(3)], axis=0)
        wh = np.concatenate([refs[4 + 3 * j].T for j in range(3)], axis=0)
        b = np.concatenate([refs[5 + 3 * j] for j in range(3)])
        return [refs[0], refs[1], refs[2], wx, wh, b]

    def _cell_fwd(self, gx, gh, prev, out, saved):
        _lib.gru_cell_fwd(gx, gh, prev[0], out[0], saved)

    def _cell_bwd(self, dh, dh_rec, carry, last, saved, prev, out, dgx, dgh):
        # carry = direct part dh (1 - u) of the next step (in), of this step (out)
        _lib.gru_cell_bwd(dh, dh_rec, None if last else carry, saved, prev[0], dgx, dgh, carry)
