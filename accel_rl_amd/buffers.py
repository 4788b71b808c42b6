"""Struct-of-arrays batch buffers resident in HBM.

Same construction-by-example API as the reference's accel_rl/buffers/batch.py
(batch_buffer / buffer_with_segs_view / view_segments / combine_distinct_buffers /
buffer_length / count_buffer_size), but every leaf is a torch tensor on the
sampler's GPU instead of a numpy / mp.RawArray array: the rollout is produced
and consumed on the device, so there is no shared-memory hand-off to mirror
(accel_rl/buffers/array.py:7-11 has no equivalent here).

Layout contract (reference: buffers/batch.py:59-76): leading dimension is
n_env * horizon, flat index = env * horizon + t; `segs_view[e]` holds views of
rows [e*horizon, (e+1)*horizon).
"""
import numpy as np
import torch

from accel_rl_amd.util.misc import struct

_TORCH_DTYPES = {
    np.dtype("bool"): torch.bool, np.dtype("uint8"): torch.uint8, np.dtype("int8"): torch.int8,
    np.dtype("int16"): torch.int16, np.dtype("int32"): torch.int32, np.dtype("int64"): torch.int64,
    np.dtype("float16"): torch.float16, np.dtype("float32"): torch.float32,
    np.dtype("float64"): torch.float64,
}


def build_array(value, length, device):
    """Zero tensor of shape (length,) + shape(value) and value's dtype."""
    if isinstance(value, torch.Tensor):
        shape, dtype = tuple(value.shape), value.dtype
    else:
        v = np.asarray(value)
        if v.dtype == object or v.dtype not in _TORCH_DTYPES:
            raise TypeError("Unsupported buffer example data type {} (nested dicts are fine, "
                            "leaves must be numeric/bool arrays)".format(v.dtype))
        shape, dtype = v.shape, _TORCH_DTYPES[v.dtype]
    return torch.zeros((length,) + tuple(shape), dtype=dtype, device=device)


def batch_buffer(example, length, device):
    if isinstance(example, dict):
        return struct(**{k: batch_buffer(v, length, device) for k, v in example.items()})
    return build_array(example, length, device)


def _walk_lengths(buf, prefix, found):
    for k, v in buf.items():
        if isinstance(v, dict):
            _walk_lengths(v, prefix + [k], found)
        else:
            found.append((prefix + [k], len(v)))


def buffer_length(buf):
    """Common leading length of every leaf (keys `segs_view` / `extra*` excluded)."""
    found = []
    _walk_lengths({k: v for k, v in buf.items()
                   if k != "segs_view" and not k.startswith("extra")}, [], found)
    lengths = {n for _, n in found}
    if len(lengths) > 1:
        raise RuntimeError("Different lengths in buffer: {}".format(found))
    return lengths.pop() if lengths else None


def _slice_tree(buf, lo, hi):
    out = struct()
    for k, v in buf.items():
        if k == "segs_view" or k.startswith("extra"):
            continue
        out[k] = _slice_tree(v, lo, hi) if isinstance(v, dict) else v[lo:hi]
    return out


def view_segments(buf, segment_length):
    length = buffer_length(buf)
    if length % segment_length != 0:
        raise ValueError("Buffer length ({}) not divisible by requested segment_length "
                         "({})".format(length, segment_length))
    return [_slice_tree(buf, lo, lo + segment_length) for lo in range(0, length, segment_length)]


def buffer_with_segs_view(examples, length, segment_length, device):
    buf = batch_buffer(examples, length, device)
    buf.segs_view = view_segments(buf, segment_length)
    return buf


def combine_distinct_buffers(buffer_1, buffer_2):
    """Merge two buffers with disjoint top-level keys (and their segment views)."""
    out = buffer_1.copy()
    other = buffer_2.copy()
    if "segs_view" in out and "segs_view" in other:
        segs = other.pop("segs_view")
        assert len(segs) == len(out.segs_view)
        for a, b in zip(out.segs_view, segs):
            a.update(b)
    out.update(other)
    return out


def count_buffer_size(buf):
    total = 0
    for k, v in buf.items():
        if k == "segs_view":
            continue
        total += count_buffer_size(v) if isinstance(v, dict) else v.numel() * v.element_size()
    return total
