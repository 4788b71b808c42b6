"""Global seeding (reference: rllab/misc/ext.py:198-220 set_seed).  Besides
python/numpy this seeds torch and keeps the RandomState that stands in for
Lasagne's private RNG (lasagne.random.set_rng), used by the conv initialiser."""
import random

import numpy as np

_layer_rng = np.random.RandomState(0)


def set_seed(seed):
    global _layer_rng
    seed = int(seed) % 4294967294
    random.seed(seed)
    np.random.seed(seed)
    _layer_rng = np.random.RandomState(seed)
    try:
        import torch
        torch.manual_seed(seed)
    except ImportError:  # pragma: no cover
        pass
    return seed


def layer_rng():
    return _layer_rng
