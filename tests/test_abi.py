"""CPU-only: the C-ABI library builds, loads, and exports exactly what
include/accel_rl_hip.h (the drop-in boundary: stateless) and include/accel_rl_hip_dev.h (development hooks) declare;
argument errors are reported without a GPU."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from accel_rl_amd import _build, _lib
    _build.build_extension()
    return _lib.load()


def _declared_functions(header="accel_rl_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(arl_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(lib):
    from accel_rl_amd import _lib
    declared = _declared_functions()
    assert len(declared) >= 14
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared
    dev = _declared_functions("accel_rl_hip_dev.h")
    assert sorted(_lib.DEV_SYMBOLS) == dev and all(n.startswith("arl_dev_") for n in dev)
    assert not set(dev) & set(declared)


def test_every_declared_symbol_is_exported(lib):
    raw = ctypes.CDLL(os.path.join(ROOT, "accel_rl_amd", "libaccel_rl_hip.so"))
    for name in _declared_functions() + _declared_functions("accel_rl_hip_dev.h"):
        assert getattr(raw, name) is not None, name


def test_the_boundary_header_keeps_no_state():
    """'no global state besides the error string'.  Every process-global switch lives in
    accel_rl_hip_dev.h; the boundary header declares none (no 'Not thread-safe' function, no mode setters), and the
    library's conv / dense entry points take their route and their co-run job as arguments."""
    text = open(os.path.join(ROOT, "include", "accel_rl_hip.h")).read()
    assert "not thread-safe" not in text.lower()
    names = _declared_functions()
    assert not [n for n in names if n.startswith("arl_dev_") or n.endswith(("_precision", "_tile_choice", "_persistent",
                                                                             "_force_generic", "_force_wave"))]
    assert "int32_t route;" in text and "arl_corun_job" in text


def test_abi_version_and_arg_errors(lib):
    assert lib.arl_abi_version() == 4
    # null pointers / bad sizes are rejected before any HIP call is made
    assert lib.arl_gae_scan(None, None, None, None, 0.99, 0.95, 4, 5, 0, None, None, None) == -1
    assert b"null" in lib.arl_last_error()
    assert lib.arl_sample_categorical(None, None, 1, 4, None, None) == -1
    assert lib.arl_valids_mask(None, 1, 1, None, None, None, None, None) == -1
    assert lib.arl_standardize(None, None, 1, 1e-6, None, None) == -1
    assert lib.arl_preprocess_frames(None, None, 1, 0, None, None) == -1
    assert lib.arl_standardize_workspace_bytes() >= 3 * 8
    assert lib.arl_env_step_served(None, None, None, None, None, None, 0, 1.0, 0.99, 0, None) == -1
    assert lib.arl_serve_conv1_supported(None, None) == 0
    assert lib.arl_conv2d_fwd_parts(None, None, None, None, None, 1, None, None, None) == -1


def test_struct_layouts_match_header(lib):
    """ctypes mirrors of the ABI structs have the C layout (sizes from gcc)."""
    import subprocess
    import tempfile
    from accel_rl_amd import _lib
    src = ('#include <stdio.h>\n#include "accel_rl_hip.h"\n#include "accel_rl_hip_dev.h"\n'
           'int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
           'sizeof(arl_game),sizeof(arl_env_state),sizeof(arl_rollout),sizeof(arl_opt_state),'
           'sizeof(arl_conv_geom),sizeof(arl_replay),sizeof(arl_fold_item),sizeof(arl_corun_job),'
           'sizeof(arl_serve_head),sizeof(arl_serve_conv1),sizeof(arl_logit_src));return 0;}')
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(_lib.ArlGame), ctypes.sizeof(_lib.ArlEnvState),
                     ctypes.sizeof(_lib.ArlRollout), ctypes.sizeof(_lib.ArlOptState),
                     ctypes.sizeof(_lib.ArlConvGeom), ctypes.sizeof(_lib.ArlReplay), ctypes.sizeof(_lib.ArlFoldItem),
                     ctypes.sizeof(_lib.ArlCorunJob), ctypes.sizeof(_lib.ArlServeHead), ctypes.sizeof(_lib.ArlServeConv1),
                     ctypes.sizeof(_lib.ArlLogitSrc)]


def test_struct_field_offsets_match_header(lib):
    """Every field of every ctypes mirror sits at the header's offsetof (a field that moves into former padding
    keeps sizeof unchanged -- arl_conv_geom.route did exactly that in round 3)."""
    import subprocess
    import tempfile
    from accel_rl_amd import _lib
    pairs = [("arl_game", _lib.ArlGame), ("arl_env_state", _lib.ArlEnvState), ("arl_rollout", _lib.ArlRollout),
             ("arl_opt_state", _lib.ArlOptState), ("arl_conv_geom", _lib.ArlConvGeom), ("arl_replay", _lib.ArlReplay),
             ("arl_fold_item", _lib.ArlFoldItem), ("arl_serve_head", _lib.ArlServeHead),
             ("arl_serve_conv1", _lib.ArlServeConv1), ("arl_dgrad_wt", _lib.ArlDgradWt),
             ("arl_logit_src", _lib.ArlLogitSrc)]
    lines = ['printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (c, f[0], c, f[0]) for c, cls in pairs for f in cls._fields_]
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "accel_rl_hip.h"\nint main(){%s return 0;}' % "\n".join(lines)
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "o.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "o")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    classes = dict(pairs)
    assert len(out) == 2 * len(lines) and len(lines) > 80
    for name, off in zip(out[0::2], out[1::2]):
        c, f = name.split(".")
        assert getattr(classes[c], f).offset == int(off), name


def test_no_cpu_fallback():
    """Host tensors are refused: the product has no CPU path."""
    import torch
    from accel_rl_amd import _lib
    with pytest.raises(RuntimeError, match="no CPU path"):
        _lib.ptr(torch.zeros(4))
