"""CNN shapes for 104x80 inputs (reference: accel_rl/policies/atari_cnn_specs.py:8-68).
Data only; spec 1 is the "NIPS large" net BASELINE config 2 names."""

cnn_specs = dict()

_SPECS = [
    # filters,            sizes,           strides,         pads,                                     hidden
    ([16, 32],            [8, 4],          [4, 2],          [(0, 0), (1, 1)],                          [256]),
    ([32, 64, 64],        [8, 4, 3],       [4, 2, 1],       [(0, 0), (1, 1), (1, 1)],                  [512]),
    ([32, 64, 64, 128, 128], [5, 3, 3, 3, 3], [3, 1, 1, 2, 1], [(0, 0), (1, 1), (1, 1), (1, 1), (1, 1)], [64, 64]),
    ([32, 64, 64, 64, 128], [4, 3, 3, 3, 3], [2, 1, 1, 1, 2], [(0, 0), (1, 1), (1, 1), (1, 1), (0, 0)], [64, 64]),
    ([16, 32, 64],        [16, 8, 4],      [3, 2, 1],       [(1, 1), (1, 2), (1, 1)],                  [256]),
]
for _i, (_f, _s, _st, _p, _h) in enumerate(_SPECS):
    cnn_specs[_i] = cnn_specs[str(_i)] = dict(
        conv_filters=_f, conv_filter_sizes=_s, conv_strides=_st, conv_pads=_p, hidden_sizes=_h)
