"""
Import shims that let the *real* reference (/root/reference, pure Python) run in
the build container, where theano / lasagne / atari_py / cv2 / path / pyprind
are absent.  Used ONLY by tests/golden/gen_golden.py (fixture generation, run
in the build container).  Nothing here is imported by the product, by the
`-m gpu` tests, by bench.py or by smoke(): /root/reference does not exist on
the GPU box.

What is real and what is a stand-in:
  real      : every accel_rl / rllab module that gets imported (the sampler,
              worker, buffers, AtariEnv, pg.util, aac_base, optimizers.util,
              runners, spaces, rllab.misc.special ...), numpy's MT19937.
  stand-ins : theano/lasagne (inert; no graph is ever executed), atari_py
              (oracle.synth_ale.SynthALE: the synthetic fixed-frame emulator
              that north_star specifies in place of ALE), cv2.resize (rounded
              2x2 box mean == OpenCV's exact-2x INTER_LINEAR/INTER_AREA result,
              see SURVEY.md a-11; parity unpinned at the cv2 boundary).
"""

import os
import sys
import types
from unittest import mock

import numpy as np

REFERENCE = "/root/reference"


class _Inert(types.ModuleType):
    """Module whose every attribute is a MagicMock (never executed for values)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        m = mock.MagicMock(name=self.__name__ + "." + name)
        setattr(self, name, m)
        return m


def _box2x_resize(src, dsize, dst=None, fx=None, fy=None, interpolation=None):
    """cv2.resize stand-in for an exact 2x decimation (see module docstring)."""
    src = np.asarray(src)
    if src.ndim == 3:
        src = src[:, :, 0]
    w, h = dsize
    assert src.shape == (2 * h, 2 * w), (src.shape, dsize)
    s = src.astype(np.uint16)
    out = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    return out.astype(np.uint8)


def install():
    if not os.path.isdir(REFERENCE):
        raise RuntimeError("reference tree not present: golden vectors can only "
                           "be regenerated in the build container")
    sys.dont_write_bytecode = True
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    here = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if here not in sys.path:
        sys.path.insert(0, here)

    # numpy >= 1.24 removed these aliases; the reference (2018) uses them
    for alias, target in (("bool", bool), ("int", int), ("float", float),
                          ("float_", np.float64), ("object", object)):
        if not hasattr(np, alias):
            setattr(np, alias, target)

    # inspect.getargspec was removed in Python 3.11 (present on 3.10); guard anyway
    import inspect
    if not hasattr(inspect, "getargspec"):
        inspect.getargspec = inspect.getfullargspec

    for name in ("theano", "theano.tensor", "theano.tensor.nnet",
                 "theano.tensor.extra_ops", "theano.gpuarray", "theano.sandbox",
                 "theano.sandbox.rng_mrg", "theano.tensor.signal",
                 "theano.tensor.signal.pool",
                 "lasagne", "lasagne.layers", "lasagne.layers.base",
                 "lasagne.init", "lasagne.nonlinearities", "lasagne.updates",
                 "lasagne.utils", "lasagne.random", "lasagne.regularization",
                 "lasagne.objectives", "path", "pyprind", "posix_ipc"):
        if name not in sys.modules:
            sys.modules[name] = _Inert(name)
    # class statements that inherit from lasagne classes need real types
    sys.modules["lasagne.init"].Initializer = type("Initializer", (), {})
    sys.modules["lasagne.layers"].Layer = type("Layer", (), {})
    sys.modules["lasagne.layers"].MergeLayer = type("MergeLayer", (), {})
    sys.modules["theano"].config = types.SimpleNamespace(floatX="float32")

    from oracle import synth_ale
    atari_py = types.ModuleType("atari_py")
    atari_py.ALEInterface = synth_ale.SynthALE
    atari_py.get_game_path = synth_ale.get_game_path
    sys.modules["atari_py"] = atari_py

    cv2 = types.ModuleType("cv2")
    cv2.INTER_NEAREST = 0
    cv2.INTER_LINEAR = 1
    cv2.resize = _box2x_resize
    sys.modules["cv2"] = cv2
