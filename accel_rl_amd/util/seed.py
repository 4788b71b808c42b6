"""Global seeding (reference: rllab/misc/ext.py:198-220 set_seed).  Besides
python/numpy this seeds torch and keeps the RandomState that stands in for
Lasagne's private RNG (lasagne.random.set_rng), used by the conv initialiser."""
import random

import numpy as np

_layer_rng = np.random.RandomState(0)


def set_seed(seed, device_generators=True):
    """device_generators=False (forked simulation workers, sampler/host_sampler.py): python, numpy, the layer RNG and
    torch's CPU generator only.  torch.manual_seed also queues a seeding call for every accelerator backend that is not
    initialised yet and records a formatted stack trace with it (torch/{cuda,xpu}/__init__.py `_lazy_call`): in a child
    forked from a process whose HIP runtime is up that walk over the source files segfaulted -- one worker start in a few
    hundred on a fresh box, every second pytest run on a used one (tests/test_host_sampler_gpu.py)."""
    global _layer_rng
    seed = int(seed) % 4294967294
    random.seed(seed)
    np.random.seed(seed)
    _layer_rng = np.random.RandomState(seed)
    try:
        import torch
        if device_generators:
            torch.manual_seed(seed)
        else:
            torch.default_generator.manual_seed(seed)
    except ImportError:  # pragma: no cover
        pass
    return seed


def layer_rng():
    return _layer_rng
