"""conv stack -> plain tanh recurrent layer -> (softmax pi, linear V)   (SURVEY 8 f3).

Mirror of the reference's AtariRnnPolicy / PgCnnRnn / RecurrentLayer
(accel_rl/policies/pg/atari_rnn_policy.py:15-180, pg/networks/pg_cnn_rnn.py:10-150,
policies/layers.py:37-106): h' = tanh(x W_xh + h W_hh + b), state key `hprev_0`, h0 = 0 (not
trainable), W_xh and W_hh drawn with NormCInit(1.0).
"""
import numpy as np

from accel_rl_amd import _lib
from accel_rl_amd.policies.atari_cnn_policy import _norm_c
from accel_rl_amd.policies.atari_lstm_policy import RecurrentCnnPolicy


class AtariRnnPolicy(RecurrentCnnPolicy):

    _gate_mult, _saved_mult, _separate_dgh = 1, 0, False
    _state_keys = ("hprev_0",)

    def _hidden_reference_init(self, fan):
        h = self._H
        self._hid_geom, self._rec_fan = [], fan
        return [_norm_c((fan, h), 1.0), _norm_c((h, h), 1.0), np.zeros(h, np.float32)], ["RnnWx", "RnnWh", "Rnnb"], h

    def _hidden_internal_shapes(self):
        h = self._H
        return [(h, self._rec_fan), (h, h), (h,)]

    def _hidden_to_reference(self, arrs):
        return [self._conv_flat_to_reference(arrs[0]), arrs[1].T, arrs[2]]

    def _hidden_to_internal(self, refs):
        return [self._conv_flat_to_internal(refs[0]), refs[1].T, refs[2]]

    def _cell_fwd(self, gx, gh, prev, out, saved):
        _lib.rnn_cell_fwd(gx, gh, out[0])

    def _cell_bwd(self, dh, dh_rec, carry, last, saved, prev, out, dgx, dgh):
        _lib.rnn_cell_bwd(dh, dh_rec, out[0], dgx)
