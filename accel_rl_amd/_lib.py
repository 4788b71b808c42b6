"""ctypes binding of libaccel_rl_hip.so (include/accel_rl_hip.h).

There is NO fallback: if the shared library is missing or a call fails, the
product raises.  PyTorch is used only as the owner of device memory and
streams; the ABI itself sees raw pointers and sizes.
"""
import ctypes as C
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libaccel_rl_hip.so")

ARL_ABI_VERSION = 4
PROMO_NEP50, PROMO_LEGACY, PROMO_ASSOC = 0, 1, 2
PPO_TIE_THEANO, PPO_TIE_MATH, PPO_TIE_BOTH = 0, 1, 2      # ARL_PPO_TIE_*: whose gradient min() / clip() hand on (accel_rl_hip.h)
OPT_ADAM, OPT_RMSPROP = 0, 1
MAX_ACTIONS = 18
REPLAY_MAX_HORIZON = 16
RAW_H, RAW_W, OBS_H, OBS_W = 210, 160, 104, 80
OPT_PARTIALS = 1024

_vp, _i32, _i64, _f32, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double


class ArlGame(C.Structure):
    _fields_ = [("bank", _vp), ("n_frames", _i32), ("n_actions", _i32),
                ("action_set", _i32 * MAX_ACTIONS), ("start_lives", _i32),
                ("life_period", _i32), ("frame_skip", _i32), ("n_stack", _i32),
                ("clip_reward", _i32), ("episodic_lives", _i32), ("resample_mode", _i32)]


RESAMPLE_BOX2X, RESAMPLE_NEAREST = 0, 1       # ARL_RESAMPLE_* (atari_env.py:155; box2x = what the reference computes)
RESAMPLE_MODES = dict(box2x=RESAMPLE_BOX2X, nearest=RESAMPLE_NEAREST)


EPOCH_WORDS = 32 * 17         # ARL_EPOCH_WORDS: launch epoch + arl_env_step's arrival tickets


class ArlEnvState(C.Structure):
    _fields_ = [("n_env", _i64), ("tick", _vp), ("emu_lives", _vp), ("env_lives", _vp),
                ("phase", _vp), ("over", _vp), ("frozen", _vp),
                ("traj_len", _vp), ("traj_nonzero", _vp), ("traj_ret", _vp),
                ("traj_raw", _vp), ("traj_disc", _vp), ("traj_curdisc", _vp),
                ("frame_a", _vp), ("frame_b", _vp), ("frame_mode", _vp), ("reset_flag", _vp),
                ("noop_ring", _vp), ("noop_cursor", _vp), ("epoch", _vp),
                ("noop_ring_len", _i32), ("envs_per_stream", _i32),
                ("done_count", _vp), ("done_int", _vp), ("done_flt", _vp),
                ("done_capacity", _i32), ("next_reset", _vp), ("launch_count", _vp)]


class ArlRollout(C.Structure):
    _fields_ = [("horizon", _i32), ("observations", _vp), ("rewards", _vp), ("dones", _vp),
                ("raw_reward", _vp), ("need_reset", _vp), ("actions", _vp), ("prob", _vp),
                ("value", _vp), ("step_obs", _vp)]


class ArlConvGeom(C.Structure):
    _fields_ = [("batch", _i64), ("in_h", _i32), ("in_w", _i32), ("in_c", _i32), ("out_c", _i32),
                ("kh", _i32), ("kw", _i32), ("stride", _i32), ("pad_h", _i32), ("pad_w", _i32), ("route", _i32)]


class ArlCorunJob(C.Structure):
    _fields_ = [("opaque", _i64 * 40)]


class ArlFoldItem(C.Structure):
    _fields_ = [("part", _vp), ("out", _vp), ("total", _i64), ("splits", _i32), ("valid", _i32)]


FOLD_MAX_ITEMS = 24


class ArlLogitSrc(C.Structure):
    _fields_ = [("part", _vp), ("bias_or_null", _vp), ("split_stride", _i64), ("splits", _i32), ("reserved", _i32)]


class ArlDgradWt(C.Structure):
    _fields_ = [("w", _vp), ("wt", _vp), ("geom", C.POINTER(ArlConvGeom))]


DGRAD_WT_MAX = 4


class ArlServeHead(C.Structure):
    _fields_ = [("hidden", ArlFoldItem), ("hidden_bias", _vp), ("hidden_relu", _i32), ("hid", _i32),
                ("w_head", _vp), ("b_head", _vp)]


class ArlServeConv1(C.Structure):
    _fields_ = [("geom", C.POINTER(ArlConvGeom)), ("w", _vp), ("bias", _vp), ("y", _vp), ("scale", _f32),
                ("relu", _i32)]


class ArlReplay(C.Structure):
    _fields_ = [("n_env", _i64), ("size", _i32), ("n_stack", _i32), ("frame_bytes", _i32),
                ("reward_horizon", _i32), ("frames", _vp), ("n_blanks", _vp), ("acts", _vp),
                ("terminals", _vp), ("rewards", _vp), ("returns", _vp)]


class ArlOptState(C.Structure):
    _fields_ = [("n_params", _i64), ("params", _vp), ("grads", _vp), ("slot0", _vp),
                ("slot1", _vp), ("step_count", _vp), ("lr_mult", _vp), ("partials", _vp),
                ("grad_norm_log", _vp), ("norm_log_len", _i32)]


_SIGNATURES = {
    "arl_abi_version": (_i32, []),
    "arl_last_error": (C.c_char_p, []),
    "arl_gae_scan": (_i32, [_vp, _vp, _vp, _vp, _f64, _f64, _i64, _i32, _i32, _vp, _vp, _vp]),
    "arl_nstep_return": (_i32, [_vp, _vp, _vp, _vp, _f64, _i64, _i32, _i32, _vp, _vp, _vp]),
    "arl_valids_mask": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "arl_standardize_workspace_bytes": (_i64, []),
    "arl_standardize": (_i32, [_vp, _vp, _i64, _f64, _vp, _vp]),
    "arl_sample_categorical": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "arl_env_act_step": (_i32, [C.POINTER(ArlGame), C.POINTER(ArlEnvState), C.POINTER(ArlRollout),
                                _vp, _vp, _vp, _vp, _i32, _i32, _f64, _f64, _vp]),
    "arl_env_frame_step": (_i32, [C.POINTER(ArlGame), C.POINTER(ArlEnvState), C.POINTER(ArlRollout),
                                  _i32, _i32, _vp]),
    "arl_env_step": (_i32, [C.POINTER(ArlGame), C.POINTER(ArlEnvState), C.POINTER(ArlRollout),
                            _vp, _vp, _vp, _vp, _i32, _i32, _f64, _f64, _i32, _i32, _vp]),
    "arl_rollout_begin": (_i32, [C.POINTER(ArlGame), C.POINTER(ArlEnvState), C.POINTER(ArlRollout), _vp]),
    "arl_env_reset": (_i32, [C.POINTER(ArlGame), C.POINTER(ArlEnvState), C.POINTER(ArlRollout),
                             _vp, _i32, _vp]),
    "arl_preprocess_frames": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "arl_copy_bytes": (_i32, [_vp, _vp, _i64, _vp]),
    "arl_ring_append": (_i32, [_vp, _i32, _vp, _i32, _vp, _vp]),
    "arl_gather_scale_obs": (_i32, [_vp, _vp, _i64, _i64, _f32, _vp, _vp]),
    "arl_gather_scale_obs_nhwc": (_i32, [_vp, _vp, _i64, _i32, _i32, _f32, _vp, _vp]),
    "arl_bias_relu": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "arl_relu_bwd_workspace_bytes": (_i64, []),
    "arl_relu_bwd_bias_grad": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    "arl_pg_head_workspace_bytes": (_i64, []),
    "arl_pg_head_infer": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "arl_pg_head_loss": (_i32, [_vp] * 11 + [_i64, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _i32] + [_vp] * 7),
    "arl_pg_head_loss_parts": (_i32, [_vp] * 11 + [_i64, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _i32] + [_vp] * 7 +
                               [_vp, _i32, _vp]),
    "arl_conv_workspace_bytes": (_i64, []),
    "arl_conv2d_fwd": (_i32, [_vp, _vp, _vp, _vp, C.POINTER(ArlConvGeom), _i32, _vp, _vp]),
    "arl_conv2d_bwd_data": (_i32, [_vp, _vp, _vp, _vp, _vp, C.POINTER(ArlConvGeom), C.POINTER(ArlCorunJob), C.POINTER(_i32), _vp]),
    "arl_conv2d_dgrad_weights": (_i32, [C.POINTER(ArlDgradWt), _i32, _vp]),
    "arl_conv2d_bwd_weight": (_i32, [_vp, _vp, _vp, C.POINTER(ArlConvGeom), _vp, _vp]),
    "arl_conv2d_bwd_weight_parts": (_i32, [_vp, _vp, _vp, C.POINTER(ArlConvGeom), _vp, _i64, C.POINTER(ArlFoldItem),
                                           _vp, C.POINTER(ArlFoldItem), _vp]),
    "arl_fold_many": (_i32, [C.POINTER(ArlFoldItem), _i32, _vp]),
    "arl_conv2d_fwd_parts": (_i32, [_vp, _vp, _vp, _vp, C.POINTER(ArlConvGeom), _i32, _vp, C.POINTER(ArlFoldItem), _vp]),
    "arl_serve_conv1_supported": (_i32, [C.POINTER(ArlGame), C.POINTER(ArlConvGeom)]),
    "arl_rollout_begin_conv1": (_i32, [C.POINTER(ArlGame), C.POINTER(ArlEnvState), C.POINTER(ArlRollout),
                                       C.POINTER(ArlServeConv1), _vp]),
    "arl_env_step_served": (_i32, [C.POINTER(ArlGame), C.POINTER(ArlEnvState), C.POINTER(ArlRollout),
                                   C.POINTER(ArlServeHead), C.POINTER(ArlServeConv1), _vp, _i32, _f64, _f64, _i32, _vp]),
    "arl_conv2d_u8_fwd": (_i32, [_vp, _i64, _vp, _f32, _vp, _vp, _vp, C.POINTER(ArlConvGeom), _i32, _vp]),
    "arl_conv2d_u8_bwd_weight_parts": (_i32, [_vp, _vp, _i64, _vp, _f32, _vp, C.POINTER(ArlConvGeom), _vp, _i64,
                                              C.POINTER(ArlFoldItem), _vp, C.POINTER(ArlFoldItem), _vp]),
    "arl_conv2d_bwd_pair": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(ArlConvGeom), _vp, _i64,
                                   C.POINTER(ArlFoldItem), _vp, C.POINTER(ArlFoldItem), C.POINTER(ArlCorunJob),
                                   C.POINTER(_i32), _vp]),
    "arl_relu_bwd_bias_parts": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, C.POINTER(ArlFoldItem), _vp]),
    "arl_replay_append": (_i32, [C.POINTER(ArlReplay), _vp, _vp, _vp, _vp, _i32, _i32, _f64, _i32, _vp]),
    "arl_replay_extract": (_i32, [C.POINTER(ArlReplay), _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "arl_sumtree_find": (_i32, [_vp, _i32, _vp, _i64, _vp, _vp]),
    "arl_sumtree_add": (_i32, [_vp, _i32, _vp, _vp, _i64, _vp]),
    "arl_sumtree_gather": (_i32, [_vp, _vp, _i64, _f64, _vp, _vp]),
    "arl_sumtree_sample": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "arl_sumtree_sample_batch": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _f64, _vp, _vp, _i32, _vp]),
    "arl_sumtree_update_pow": (_i32, [_vp, _i32, _vp, _vp, _vp, _f64, _i64, _vp]),
    "arl_is_weights": (_i32, [_vp, _i64, _f64, _vp, _vp]),
    "arl_priority_diffs": (_i32, [_vp, _vp, _i64, _f64, _vp, _vp]),
    "arl_catdqn_act": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "arl_catdqn_loss": (_i32, [_vp] * 8 + [_i64, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _vp, _vp, _vp, _vp]),
    "arl_catdqn_loss_parts": (_i32, [_vp] * 8 + [_i64, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _i32,
                               _vp]),
    "arl_dqn_act": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "arl_dqn_loss": (_i32, [_vp] * 7 + [_i64, _i32, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _vp]),
    "arl_lstm_cell_fwd": (_i32, [_vp, _i64, _vp, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "arl_lstm_cell_bwd": (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _vp]),
    "arl_gru_cell_fwd": (_i32, [_vp, _i64, _vp, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _i64, _vp]),
    "arl_gru_cell_bwd": (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _vp]),
    "arl_rnn_cell_fwd": (_i32, [_vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp]),
    "arl_rnn_cell_bwd": (_i32, [_vp, _i64, _vp, _vp, _i64, _i64, _i32, _vp, _i64, _vp]),
    "arl_opt_step": (_i32, [C.POINTER(ArlOptState), _i32, _f32, _f32, _f32, _f32, _f32, _f32, _vp]),
    "arl_opt_step_noclip": (_i32, [C.POINTER(ArlOptState), _i32, _f32, _f32, _f32, _f32, _f32, _i32, _vp, _vp, _vp]),
    "arl_opt_finish": (_i32, [C.POINTER(ArlOptState), _i32, _f32, _vp, _vp, _vp]),
    "arl_opt_step_noclip_split": (_i32, [C.POINTER(ArlOptState), _i32, _f32, _f32, _f32, _f32, _f32, _i32, _vp, _vp,
                                         _i64, _i64, _i32, _vp]),
    "arl_opt_finish_split": (_i32, [C.POINTER(ArlOptState), _i32, _f32, _vp, _vp, _i64, _vp]),
    "arl_corun_job_init": (_i32, [C.POINTER(ArlCorunJob), C.POINTER(ArlOptState), _i32, _f32, _f32, _f32, _f32, _f32,
                                  _i32, _vp, _vp, _i64, _i64]),
    "arl_corun_job_run": (_i32, [C.POINTER(ArlCorunJob), _vp]),
}

# include/accel_rl_hip_dev.h: development hooks (tests / tools), not part of the drop-in boundary
_DEV_SIGNATURES = {
    "arl_dev_conv_trace_buffer": (None, [_vp]),
    "arl_dev_conv_force_generic": (None, [_i32]),
    "arl_dev_conv_variant": (None, [_i32]),
    "arl_dev_fwd_tile": (None, [_i32]),
    "arl_dev_dgrad_wt": (None, [_i32]),
    "arl_dev_fold_wide_from": (None, [_i32]),
    "arl_dev_scan_force_wave": (None, [_i32]),
    "arl_dev_scan_wave_groups": (None, [_i32]),
    "arl_dev_env_variant": (None, [_i32]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)
DEV_SYMBOLS = tuple(_DEV_SIGNATURES)

# arl_conv_geom.route (ARL_CONV_ROUTE_*): how the fp32 contractions are computed.  The library keeps no mode; this
# module's default is what conv_geom() / dense_geom() stamp into the geometries they build (host-side policy).
ROUTE_SPLIT9, ROUTE_FP32, ROUTE_SPLIT6, ROUTE_BF16 = 0, 1, 6, 2
_PRECISION_TO_ROUTE = {9: ROUTE_SPLIT9, 0: ROUTE_FP32, 6: ROUTE_SPLIT6, 1: ROUTE_BF16}
default_route = _PRECISION_TO_ROUTE[int(os.environ.get("ARL_CONV_PRECISION", "9"))]    # measurement switch (tools/, bench A/B)


def set_conv_precision(mode):
    """Route of the geometries built from now on: 9 = nine exact bf16-split products (the default), 6 = six,
    0 = the fp32 MFMA chain, 1 = plain bf16 operands (ARL_CONV_ROUTE_BF16: the labelled reduced-precision option, not an
    fp32 contraction).  Geometries already built keep theirs (ArlConvGeom.route)."""
    global default_route
    if mode not in _PRECISION_TO_ROUTE:
        raise ValueError("conv precision: 0 (fp32 MFMA), 6 or 9 (bf16-split products) or 1 (bf16 operands)")
    default_route = _PRECISION_TO_ROUTE[mode]


def conv_precision():
    return {v: k for k, v in _PRECISION_TO_ROUTE.items()}[default_route]

_lib = None


def load():
    """Load (once) and type the shared library.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libaccel_rl_hip.so is not built (%s).  Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `python accel_rl_amd/_build.py`.  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in list(_SIGNATURES.items()) + list(_DEV_SIGNATURES.items()):
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.arl_abi_version() != ARL_ABI_VERSION:
        raise RuntimeError("libaccel_rl_hip.so ABI %d != binding %d" % (lib.arl_abi_version(), ARL_ABI_VERSION))
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        msg = load().arl_last_error().decode(errors="replace")
        raise RuntimeError("%s failed (code %d): %s" % (what, rc, msg))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("accel_rl_amd kernels need device (HIP) tensors; got a %s tensor -- "
                           "there is no CPU path" % t.device)
    if not t.is_contiguous():
        raise RuntimeError("tensor must be contiguous")
    return t.data_ptr()


def staged_ptr(t):
    """Pointer of a device tensor OR of a pinned host tensor (the device addresses pinned memory directly)."""
    if not (t.is_cuda or t.is_pinned()):
        raise RuntimeError("staging copies take device tensors or PINNED host tensors; got pageable host memory")
    if not t.is_contiguous():
        raise RuntimeError("tensor must be contiguous")
    return t.data_ptr()


STAGE_WITH_KERNELS = os.environ.get("ARL_STAGE_KERNELS", "1") != "0"      # (A/B switch: 0 = framework copies = memcpy nodes)


def copy_bytes(dst, src, stream=None):
    """dst <- src as a kernel (a kernel node when captured); either side may be a pinned host tensor."""
    if not STAGE_WITH_KERNELS:
        dst.view(-1).copy_(src.view(-1), non_blocking=True)
        return
    n = src.numel() * src.element_size()
    assert n == dst.numel() * dst.element_size(), "size mismatch"
    _check(load().arl_copy_bytes(staged_ptr(dst), staged_ptr(src), n, stream_ptr(stream)), "arl_copy_bytes")


def ring_append(src, ring, counter, stream=None):
    """ring[counter % len(ring)] = src; counter += 1 (device-side; see arl_ring_append)."""
    _want(src, torch.float32, "src")
    _want(ring, torch.float32, "ring")
    _want(counter, torch.int32, "counter")
    if not (ring.is_contiguous() and src.is_contiguous() and ring.shape[0] > 0 and ring[0].numel() == src.numel()):
        raise ValueError("ring_append: a ring slot must hold exactly the %d floats of src (ring %s)" %
                         (src.numel(), tuple(ring.shape)))
    _check(load().arl_ring_append(ptr(src), src.numel(), ptr(ring), ring.shape[0], ptr(counter), stream_ptr(stream)),
           "arl_ring_append")


def stream_ptr(stream=None):
    s = torch.cuda.current_stream() if stream is None else stream
    return s.cuda_stream


def _want(t, dtype, name):
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))


# ---------------------------------------------------------------------------
# thin typed wrappers (device tensors in, device tensors out)
# ---------------------------------------------------------------------------

def gae_scan(rewards, values, dones, last_values, discount, gae_lambda, n_env, horizon,
             advantages, returns, promo=PROMO_NEP50, stream=None):
    for t, n in ((rewards, "rewards"), (values, "values"), (last_values, "last_values"),
                 (advantages, "advantages"), (returns, "returns")):
        _want(t, torch.float32, n)
    if dones.dtype not in (torch.uint8, torch.bool):
        raise TypeError("dones must be uint8/bool")
    assert rewards.numel() == values.numel() == dones.numel() == n_env * horizon
    assert last_values.numel() == n_env and advantages.numel() == returns.numel() == n_env * horizon
    _check(load().arl_gae_scan(ptr(rewards), ptr(values), ptr(dones), ptr(last_values),
                               float(discount), float(gae_lambda), n_env, horizon, promo,
                               ptr(advantages), ptr(returns), stream_ptr(stream)), "arl_gae_scan")


def nstep_return(rewards, dones, values, last_values, discount, n_env, horizon,
                 returns, advantages, promo=PROMO_NEP50, stream=None):
    for t, n in ((rewards, "rewards"), (values, "values"), (last_values, "last_values"),
                 (advantages, "advantages"), (returns, "returns")):
        _want(t, torch.float32, n)
    assert rewards.numel() == values.numel() == dones.numel() == n_env * horizon
    assert last_values.numel() == n_env and advantages.numel() == returns.numel() == n_env * horizon
    _check(load().arl_nstep_return(ptr(rewards), ptr(dones), ptr(values), ptr(last_values),
                                   float(discount), n_env, horizon, promo, ptr(returns),
                                   ptr(advantages), stream_ptr(stream)), "arl_nstep_return")


def valids_mask(reset_flags, n_env, horizon, valids, advantages=None, returns=None, values=None,
                stream=None):
    _want(valids, torch.int8, "valids")
    assert reset_flags.numel() == valids.numel() == n_env * horizon
    _check(load().arl_valids_mask(ptr(reset_flags), n_env, horizon, ptr(valids), ptr(advantages),
                                  ptr(returns), ptr(values), stream_ptr(stream)), "arl_valids_mask")


def standardize_workspace(device):
    return torch.empty(load().arl_standardize_workspace_bytes() // 8, dtype=torch.float64, device=device)


def standardize(advantages, valids, workspace, eps=1e-6, stream=None):
    _want(advantages, torch.float32, "advantages")
    if valids is not None:
        _want(valids, torch.int8, "valids")
        assert valids.numel() == advantages.numel()
    _check(load().arl_standardize(ptr(advantages), ptr(valids), advantages.numel(), float(eps),
                                  ptr(workspace), stream_ptr(stream)), "arl_standardize")


def sample_categorical(prob, uniforms, actions, stream=None):
    _want(prob, torch.float32, "prob")
    _want(uniforms, torch.float64, "uniforms")
    _want(actions, torch.uint8, "actions")
    batch, n_act = prob.shape
    assert uniforms.numel() == batch and actions.numel() == batch
    _check(load().arl_sample_categorical(ptr(prob), ptr(uniforms), batch, n_act, ptr(actions),
                                         stream_ptr(stream)), "arl_sample_categorical")


def preprocess_frames(raw_a, raw_b, out, stream=None, resample="box2x"):
    _want(raw_b, torch.uint8, "raw_b")
    n = raw_b.shape[0]
    assert tuple(raw_b.shape[1:3]) == (RAW_H, RAW_W) and tuple(out.shape) == (n, OBS_H, OBS_W)
    _check(load().arl_preprocess_frames(ptr(raw_a), ptr(raw_b), n, RESAMPLE_MODES[resample], ptr(out),
                                        stream_ptr(stream)), "arl_preprocess_frames")


def gather_scale_obs(obs, idx, out, scale, stream=None):
    _want(obs, torch.uint8, "obs")
    _want(out, torch.float32, "out")
    if idx is not None:
        _want(idx, torch.int32, "idx")
    batch = out.shape[0]
    row_bytes = obs[0].numel()
    assert out[0].numel() == row_bytes and (idx is None or idx.numel() == batch)
    _check(load().arl_gather_scale_obs(ptr(obs), ptr(idx), batch, row_bytes, float(scale), ptr(out),
                                       stream_ptr(stream)), "arl_gather_scale_obs")


def env_act_step(game, state, rollout, prob, value, uniforms, step, mid_batch_reset,
                 max_path_length, discount, active=None, stream=None):
    _check(load().arl_env_act_step(C.byref(game), C.byref(state), C.byref(rollout), ptr(prob),
                                   ptr(value), ptr(uniforms), ptr(active), step,
                                   int(bool(mid_batch_reset)),
                                   float(max_path_length), float(discount), stream_ptr(stream)),
           "arl_env_act_step")


def env_step(game, state, rollout, prob, value, uniforms, step, mid_batch_reset, max_path_length, discount,
             max_start_noops, active=None, single_write=False, stream=None):
    """act_step + frame_step (+ epoch bump) as ONE launch."""
    _check(load().arl_env_step(C.byref(game), C.byref(state), C.byref(rollout), ptr(prob), ptr(value),
                               ptr(uniforms), ptr(active), step, int(bool(mid_batch_reset)),
                               float(max_path_length), float(discount), int(max_start_noops),
                               int(bool(single_write)), stream_ptr(stream)), "arl_env_step")


def serve_conv1_supported(game, geom):
    """Can arl_env_step_served evaluate this first convolution inside its launch?"""
    return bool(load().arl_serve_conv1_supported(C.byref(game), C.byref(geom)))


def rollout_begin_conv1(game, state, rollout, conv1, stream=None):
    """rollout_begin + the first convolution of the rows it copies (conv1: ArlServeConv1), one launch."""
    _check(load().arl_rollout_begin_conv1(C.byref(game), C.byref(state), C.byref(rollout), C.byref(conv1),
                                          stream_ptr(stream)), "arl_rollout_begin_conv1")


def env_step_served(game, state, rollout, head, conv1, uniforms, step, max_path_length, discount, max_start_noops,
                    stream=None):
    """Hidden-layer fold + heads + softmax + action draw + env step (+ the next observation's conv 1) as ONE launch.
    head: ArlServeHead; conv1: ArlServeConv1 or None.  Both hold raw pointers: the caller keeps the tensors alive."""
    _want(uniforms, torch.float64, "uniforms")
    _check(load().arl_env_step_served(C.byref(game), C.byref(state), C.byref(rollout), C.byref(head),
                                      None if conv1 is None else C.byref(conv1), ptr(uniforms), int(step),
                                      float(max_path_length), float(discount), int(max_start_noops),
                                      stream_ptr(stream)), "arl_env_step_served")


def env_frame_step(game, state, rollout, step, max_start_noops, stream=None):
    _check(load().arl_env_frame_step(C.byref(game), C.byref(state), C.byref(rollout), step,
                                     int(max_start_noops), stream_ptr(stream)), "arl_env_frame_step")


def rollout_begin(game, state, rollout, stream=None):
    """observations[:, 0] = step_obs and done_count = 0, one launch."""
    _check(load().arl_rollout_begin(C.byref(game), C.byref(state), C.byref(rollout), stream_ptr(stream)),
           "arl_rollout_begin")


def env_reset(game, state, rollout, flags, max_start_noops, stream=None):
    _check(load().arl_env_reset(C.byref(game), C.byref(state), C.byref(rollout), ptr(flags),
                                int(max_start_noops), stream_ptr(stream)), "arl_env_reset")


def opt_step(opt, method, learning_rate, avg_factor, clip, beta1_or_rho, beta2, epsilon, stream=None):
    _check(load().arl_opt_step(C.byref(opt), method, learning_rate, avg_factor,
                               0.0 if clip is None else clip, beta1_or_rho, beta2, epsilon,
                               stream_ptr(stream)), "arl_opt_step")


OPT_NORM_SLOTS, OPT_NORM_BLOCKS = 64, 2048


def opt_step_noclip(opt, method, learning_rate, avg_factor, beta1_or_rho, beta2, epsilon, k, step_pp, norm_parts,
                    stream=None):
    _check(load().arl_opt_step_noclip(C.byref(opt), method, learning_rate, avg_factor, beta1_or_rho, beta2, epsilon,
                                      int(k), ptr(step_pp), ptr(norm_parts), stream_ptr(stream)), "arl_opt_step_noclip")


def opt_finish(opt, n_updates, avg_factor, step_pp, norm_parts, stream=None, hole_count=0):
    """Close a call of no-clip updates; hole_count: the call's updates were split around a hole of that size."""
    _check(load().arl_opt_finish_split(C.byref(opt), int(n_updates), avg_factor, ptr(step_pp), ptr(norm_parts),
                                       int(hole_count), stream_ptr(stream)), "arl_opt_finish_split")


def opt_step_noclip_split(opt, method, learning_rate, avg_factor, beta1_or_rho, beta2, epsilon, k, step_pp,
                          norm_parts, hole_first, hole_count, part, stream=None):
    """part 0: the no-clip update of everything but [hole_first, hole_first + hole_count); part 1: of that range."""
    _check(load().arl_opt_step_noclip_split(C.byref(opt), method, learning_rate, avg_factor, beta1_or_rho, beta2,
                                            epsilon, int(k), ptr(step_pp), ptr(norm_parts), int(hole_first),
                                            int(hole_count), int(part), stream_ptr(stream)),
           "arl_opt_step_noclip_split")


def corun_job(opt, method, learning_rate, avg_factor, beta1_or_rho, beta2, epsilon, k, step_pp, norm_parts,
              hole_first, hole_count):
    """Part 1 of update k (the hole) as a job for a data-gradient launch to carry (arl_corun_job): plain data, held
    by the caller; conv2d_bwd_data / FoldList.conv2d_bwd_pair take it through `corun=`, corun_job_run runs it alone."""
    job = ArlCorunJob()
    _check(load().arl_corun_job_init(C.byref(job), C.byref(opt), method, learning_rate, avg_factor, beta1_or_rho, beta2,
                                     epsilon, int(k), ptr(step_pp), ptr(norm_parts), int(hole_first), int(hole_count)),
           "arl_corun_job_init")
    return job


def corun_job_run(job, stream=None):
    _check(load().arl_corun_job_run(C.byref(job), stream_ptr(stream)), "arl_corun_job_run")


# ---------------------------------------------------------------------------
# learner glue (csrc/learner.hip)
# ---------------------------------------------------------------------------

def gather_scale_obs_nhwc(obs, idx, out, scale, stream=None):
    """obs u8[n,4,H,W] -> out f32 with NHWC memory ([B,H,W,4] contiguous)."""
    _want(obs, torch.uint8, "obs")
    _want(out, torch.float32, "out")
    if idx is not None:
        _want(idx, torch.int32, "idx")
    batch = out.shape[0]
    channels, plane = obs.shape[1], obs.shape[2] * obs.shape[3]
    assert out.numel() == batch * channels * plane
    _check(load().arl_gather_scale_obs_nhwc(ptr(obs), ptr(idx), batch, channels, plane, float(scale),
                                            out.data_ptr(), stream_ptr(stream)),
           "arl_gather_scale_obs_nhwc")


def bias_relu(x, bias, rows, channels, stream=None):
    _check(load().arl_bias_relu(x.data_ptr(), bias.data_ptr(), rows, channels, stream_ptr(stream)),
           "arl_bias_relu")


def relu_bwd_workspace(device):
    return torch.empty(load().arl_relu_bwd_workspace_bytes() // 4, dtype=torch.float32, device=device)


def relu_bwd_bias_grad(dy, y, rows, channels, dbias, workspace, stream=None):
    _check(load().arl_relu_bwd_bias_grad(dy.data_ptr(), y.data_ptr(), rows, channels,
                                         dbias.data_ptr(), ptr(workspace), stream_ptr(stream)),
           "arl_relu_bwd_bias_grad")


def pg_head_workspace(device):
    return torch.zeros(load().arl_pg_head_workspace_bytes() // 4, dtype=torch.float32, device=device)


def pg_head_infer(h, w_head, b_head, prob, value, stream=None):
    batch, hid = h.shape
    n_act = prob.shape[1]
    _check(load().arl_pg_head_infer(ptr(h), w_head.data_ptr(), b_head.data_ptr(), batch, hid, n_act,
                                    ptr(prob), ptr(value), stream_ptr(stream)), "arl_pg_head_infer")


def pg_head_loss(h, w_head, b_head, actions, advantages, returns, old_prob, valids, idx, lr_mult,
                 inv_count, n_actions, kind, clip_param, v_loss_coeff, ent_loss_coeff,
                 dout, dh, dw_head, db_head, loss4, workspace, stream=None, relu_mask_dh=False,
                 tie_rule=PPO_TIE_THEANO):
    """relu_mask_dh: h is a rectifier's output; return dh already multiplied by (h > 0)."""
    batch, hid = h.shape
    _check(load().arl_pg_head_loss(
        ptr(h), w_head.data_ptr(), b_head.data_ptr(), ptr(actions), ptr(advantages), ptr(returns),
        ptr(old_prob), ptr(valids), ptr(idx), ptr(lr_mult), ptr(inv_count), batch, hid, n_actions,
        kind, int(tie_rule), float(clip_param), float(v_loss_coeff), float(ent_loss_coeff), int(bool(relu_mask_dh)), ptr(dout), ptr(dh),
        dw_head.data_ptr(), db_head.data_ptr(), ptr(loss4), ptr(workspace), stream_ptr(stream)),
        "arl_pg_head_loss")


# ---------------------------------------------------------------------------
# fp32 MFMA contractions (csrc/mfma_conv.hip); activations NHWC, weights (K, kh, kw, C)
# ---------------------------------------------------------------------------

def conv_geom(batch, in_h, in_w, in_c, out_c, kh, kw, stride, pad_h, pad_w, route=None):
    return ArlConvGeom(batch, in_h, in_w, in_c, out_c, kh, kw, stride, pad_h, pad_w,
                       default_route if route is None else route)


def dense_geom(batch, fan_in, units, route=None):
    return ArlConvGeom(batch, 1, 1, fan_in, units, 1, 1, 1, 0, 0, default_route if route is None else route)


def with_route(geom, route=None):
    """A copy of geom on another route (default: this module's current default)."""
    g = ArlConvGeom.from_buffer_copy(geom)
    g.route = default_route if route is None else route
    return g


def conv_out_hw(g):
    return ((g.in_h + 2 * g.pad_h - g.kh) // g.stride + 1, (g.in_w + 2 * g.pad_w - g.kw) // g.stride + 1)


def conv_workspace(device):
    return torch.empty(load().arl_conv_workspace_bytes() // 4, dtype=torch.float32, device=device)


def conv2d_fwd(x, w, bias, y, geom, relu, workspace, stream=None):
    """y[B,Ho,Wo,K] = act(conv(x[B,H,W,C], w[K,kh,kw,C]) + bias); all fp32 contiguous memory."""
    for t, n in ((x, "x"), (w, "w"), (y, "y")):
        _want(t, torch.float32, n)
    ho, wo = conv_out_hw(geom)
    assert x.numel() == geom.batch * geom.in_h * geom.in_w * geom.in_c, "x size"
    assert w.numel() == geom.out_c * geom.kh * geom.kw * geom.in_c, "w size"
    assert y.numel() == geom.batch * ho * wo * geom.out_c, "y size"
    _check(load().arl_conv2d_fwd(x.data_ptr(), w.data_ptr(), None if bias is None else bias.data_ptr(),
                                 y.data_ptr(), C.byref(geom), int(bool(relu)), ptr(workspace),
                                 stream_ptr(stream)), "arl_conv2d_fwd")


def conv2d_fwd_parts(x, w, bias, y, geom, relu, workspace, stream=None):
    """conv2d_fwd that leaves a split reduction unfolded: returns the ArlFoldItem describing the partial sums in
    `workspace` (splits == 0: the launch did not split and y is final, bias / rectifier applied)."""
    for t, n in ((x, "x"), (w, "w"), (y, "y")):
        _want(t, torch.float32, n)
    ho, wo = conv_out_hw(geom)
    assert x.numel() == geom.batch * geom.in_h * geom.in_w * geom.in_c, "x size"
    assert w.numel() == geom.out_c * geom.kh * geom.kw * geom.in_c, "w size"
    assert y.numel() == geom.batch * ho * wo * geom.out_c, "y size"
    item = ArlFoldItem()
    _check(load().arl_conv2d_fwd_parts(x.data_ptr(), w.data_ptr(), None if bias is None else bias.data_ptr(),
                                       y.data_ptr(), C.byref(geom), int(bool(relu)), ptr(workspace), C.byref(item),
                                       stream_ptr(stream)), "arl_conv2d_fwd_parts")
    return item


def conv2d_u8_supported(in_h, in_w, out_c, kh, kw, stride, pad_h, pad_w):
    """Geometries arl_conv2d_u8_fwd / _bwd_weight_parts accept (see include/accel_rl_hip.h)."""
    return (pad_h == 0 and pad_w == 0 and stride % 4 == 0 and in_w % 4 == 0 and (in_h * in_w) % 4 == 0 and
            kw in (4, 8, 16) and kh % (16 // kw) == 0 and out_c % 4 == 0 and out_c <= 32)


def _u8_rows(obs, idx, geom):
    _want(obs, torch.uint8, "obs")
    assert obs.is_contiguous() and obs.numel() == obs.shape[0] * geom.in_c * geom.in_h * geom.in_w, "obs shape"
    if idx is not None:
        _want(idx, torch.int32, "idx")
        assert idx.numel() == geom.batch, "idx length"
    else:
        assert obs.shape[0] == geom.batch, "obs rows"


def conv2d_u8_fwd(obs, idx, scale, w, bias, y, geom, relu, stream=None):
    """y[B,Ho,Wo,K] = act(conv(float(obs[idx]) * scale, w[K,C,kh,kw]) + bias), obs u8[n,C,H,W] read in place."""
    _u8_rows(obs, idx, geom)
    ho, wo = conv_out_hw(geom)
    assert w.numel() == geom.out_c * geom.kh * geom.kw * geom.in_c, "w size"
    assert y.numel() == geom.batch * ho * wo * geom.out_c, "y size"
    _check(load().arl_conv2d_u8_fwd(obs.data_ptr(), obs.shape[0], ptr(idx), float(scale), w.data_ptr(), ptr(bias),
                                    y.data_ptr(), C.byref(geom), int(bool(relu)), stream_ptr(stream)),
           "arl_conv2d_u8_fwd")


def conv2d_dgrad_weights(layers, stream=None):
    """layers: [(w, wt, geom)] -- wt <- the data gradient's k-contiguous copy of w (arl_conv2d_dgrad_weights), one launch."""
    assert 0 < len(layers) <= DGRAD_WT_MAX
    items = (ArlDgradWt * len(layers))()
    for it, (w, wt, geom) in zip(items, layers):
        _want(w, torch.float32, "w")
        _want(wt, torch.float32, "wt")
        assert w.numel() == wt.numel() == geom.out_c * geom.kh * geom.kw * geom.in_c, "w / wt size"
        it.w, it.wt, it.geom = ptr(w), ptr(wt), C.pointer(geom)
    _check(load().arl_conv2d_dgrad_weights(items, len(layers), stream_ptr(stream)), "arl_conv2d_dgrad_weights")


def conv2d_bwd_data(dy, w, mask, dx, geom, stream=None, corun=None, wt=None):
    """corun: an ArlCorunJob the launch may carry; returns True if it did (else the caller runs it: corun_job_run).
    wt: conv2d_dgrad_weights' copy of w (same results; the split kernels read it instead of w)."""
    ho, wo = conv_out_hw(geom)
    assert dy.numel() == geom.batch * ho * wo * geom.out_c, "dy size"
    assert dx.numel() == geom.batch * geom.in_h * geom.in_w * geom.in_c, "dx size"
    assert mask is None or mask.numel() == dx.numel()
    taken = _i32(0)
    _check(load().arl_conv2d_bwd_data(dy.data_ptr(), w.data_ptr(), ptr(wt), None if mask is None else mask.data_ptr(),
                                      dx.data_ptr(), C.byref(geom), None if corun is None else C.byref(corun),
                                      C.byref(taken), stream_ptr(stream)), "arl_conv2d_bwd_data")
    return bool(taken.value)


def conv2d_bwd_weight(dy, x, dw, geom, workspace, stream=None):
    ho, wo = conv_out_hw(geom)
    assert dy.numel() == geom.batch * ho * wo * geom.out_c, "dy size"
    assert x.numel() == geom.batch * geom.in_h * geom.in_w * geom.in_c, "x size"
    assert dw.numel() == geom.out_c * geom.kh * geom.kw * geom.in_c, "dw size"
    _check(load().arl_conv2d_bwd_weight(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), C.byref(geom),
                                        ptr(workspace), stream_ptr(stream)), "arl_conv2d_bwd_weight")


class FoldList(object):
    """Pending split folds of one backward pass (arl_fold_item): the *_parts calls append, run()
    folds everything in one launch.  Every appended call needs its own workspace tensor."""

    def __init__(self):
        self._items = (ArlFoldItem * FOLD_MAX_ITEMS)()
        self._n = 0
        self.last_dw_in_place = False
        self.corun_taken = False

    def _next(self):
        assert self._n < FOLD_MAX_ITEMS, "too many pending folds"
        self._n += 1
        return C.byref(self._items[self._n - 1])

    def _bias_slot(self, dbias):
        """(dbias pointer, item pointer) for the optional bias-gradient partials of a weight-gradient call."""
        if dbias is None:
            return None, None
        return dbias.data_ptr(), self._next()

    def _bias_done(self, dbias):
        """False when the kernel could not produce the bias partials (generic path): drop the slot."""
        if dbias is None:
            return True
        if self._items[self._n - 1].splits < 0:
            self._n -= 1
            return False
        return True

    def conv2d_bwd_weight(self, dy, x, dw, geom, workspace, dbias=None, stream=None):
        """dw partials (and, with dbias, the column sums of dy) for the deferred fold.  Returns False when
        dbias was asked for but not produced (the caller then uses relu_bwd_bias_grad)."""
        ho, wo = conv_out_hw(geom)
        assert dy.numel() == geom.batch * ho * wo * geom.out_c, "dy size"
        assert x.numel() == geom.batch * geom.in_h * geom.in_w * geom.in_c, "x size"
        assert dw.numel() == geom.out_c * geom.kh * geom.kw * geom.in_c, "dw size"
        item = self._next()
        pb, ib = self._bias_slot(dbias)
        _check(load().arl_conv2d_bwd_weight_parts(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), C.byref(geom),
                                                  ptr(workspace), workspace.numel() * workspace.element_size(),
                                                  item, pb, ib, stream_ptr(stream)), "arl_conv2d_bwd_weight_parts")
        return self._bias_done(dbias)

    def conv2d_u8_bwd_weight(self, dy, obs, idx, scale, dw, geom, workspace, dbias=None, stream=None):
        """dw[K,C,kh,kw] partials (and, with dbias, the column sums of dy) of a convolution on u8 observations."""
        _u8_rows(obs, idx, geom)
        ho, wo = conv_out_hw(geom)
        assert dy.numel() == geom.batch * ho * wo * geom.out_c, "dy size"
        assert dw.numel() == geom.out_c * geom.kh * geom.kw * geom.in_c, "dw size"
        item = self._next()
        pb, ib = self._bias_slot(dbias)
        _check(load().arl_conv2d_u8_bwd_weight_parts(dy.data_ptr(), obs.data_ptr(), obs.shape[0], ptr(idx),
                                                     float(scale), dw.data_ptr(), C.byref(geom), ptr(workspace),
                                                     workspace.numel() * workspace.element_size(), item, pb, ib,
                                                     stream_ptr(stream)), "arl_conv2d_u8_bwd_weight_parts")
        return self._bias_done(dbias)

    def conv2d_bwd_pair(self, dy, w, mask, dx, x, dw, geom, workspace, dbias=None, stream=None, corun=None, wt=None):
        """dx (times mask > 0 if given) and (deferred) dw [+ dbias] of one layer in a single launch.
        corun: an ArlCorunJob the data-gradient launch may carry; `self.corun_taken` says whether it did.
        wt: conv2d_dgrad_weights' copy of w for the data gradient."""
        ho, wo = conv_out_hw(geom)
        assert dy.numel() == geom.batch * ho * wo * geom.out_c, "dy size"
        assert x.numel() == dx.numel() == geom.batch * geom.in_h * geom.in_w * geom.in_c, "x / dx size"
        assert dw.numel() == w.numel() == geom.out_c * geom.kh * geom.kw * geom.in_c, "w / dw size"
        item = self._next()
        slot = self._n - 1
        pb, ib = self._bias_slot(dbias)
        taken = _i32(0)
        _check(load().arl_conv2d_bwd_pair(dy.data_ptr(), w.data_ptr(), ptr(wt), ptr(mask), dx.data_ptr(), x.data_ptr(),
                                          dw.data_ptr(), C.byref(geom), ptr(workspace),
                                          workspace.numel() * workspace.element_size(), item, pb, ib,
                                          None if corun is None else C.byref(corun), C.byref(taken),
                                          stream_ptr(stream)), "arl_conv2d_bwd_pair")
        self.corun_taken = bool(taken.value)
        self.last_dw_in_place = self._items[slot].splits == 0      # no split partials: dw is final as written
        return self._bias_done(dbias)

    def relu_bwd_bias_grad(self, dy, y, rows, channels, dbias, workspace, stream=None):
        _check(load().arl_relu_bwd_bias_parts(dy.data_ptr(), y.data_ptr(), rows, channels, dbias.data_ptr(),
                                              ptr(workspace), self._next(), stream_ptr(stream)),
               "arl_relu_bwd_bias_parts")

    def pg_head_loss(self, h, w_head, b_head, actions, advantages, returns, old_prob, valids, idx, lr_mult,
                     inv_count, n_actions, kind, clip_param, v_loss_coeff, ent_loss_coeff,
                     dout, dh, dw_head, db_head, loss4, workspace, stream=None, relu_mask_dh=False,
                 tie_rule=PPO_TIE_THEANO, dgrad_weights=None):
        """pg_head_loss with its three small folds (dw_head, db_head, loss4) left to run().
        dgrad_weights: [(w, wt, geom)] -- the launch also writes these layers' k-contiguous weight copies
        (conv2d_dgrad_weights' work in extra workgroups: one launch less per backward pass)."""
        batch, hid = h.shape
        assert self._n + 3 <= FOLD_MAX_ITEMS, "too many pending folds"
        first = C.byref(self._items[self._n])
        self._n += 3
        wt_items, n_wt = None, 0
        if dgrad_weights:
            n_wt = len(dgrad_weights)
            assert n_wt <= DGRAD_WT_MAX
            wt_items = (ArlDgradWt * n_wt)()
            for it, (w, wt, geom) in zip(wt_items, dgrad_weights):
                it.w, it.wt, it.geom = ptr(w), ptr(wt), C.pointer(geom)
        _check(load().arl_pg_head_loss_parts(
            ptr(h), w_head.data_ptr(), b_head.data_ptr(), ptr(actions), ptr(advantages), ptr(returns),
            ptr(old_prob), ptr(valids), ptr(idx), ptr(lr_mult), ptr(inv_count), batch, hid, n_actions,
            kind, int(tie_rule), float(clip_param), float(v_loss_coeff), float(ent_loss_coeff), int(bool(relu_mask_dh)), ptr(dout),
            ptr(dh), dw_head.data_ptr(), db_head.data_ptr(), ptr(loss4), ptr(workspace), first,
            None if wt_items is None else C.cast(wt_items, _vp), n_wt, stream_ptr(stream)),
            "arl_pg_head_loss_parts")

    def run(self, stream=None):
        n, self._n = self._n, 0
        _check(load().arl_fold_many(self._items, n, stream_ptr(stream)), "arl_fold_many")


# ---------------------------------------------------------------------------
# replay memory + sum tree (csrc/replay.hip)
# ---------------------------------------------------------------------------

def replay_append(rb, observations, actions, rewards, dones, horizon, idx, discount, promo=PROMO_NEP50,
                  stream=None):
    _want(observations, torch.uint8, "observations")
    _want(actions, torch.uint8, "actions")
    _want(rewards, torch.float32, "rewards")
    _check(load().arl_replay_append(C.byref(rb), ptr(observations), ptr(actions), ptr(rewards), ptr(dones),
                                    horizon, idx, float(discount), promo, stream_ptr(stream)), "arl_replay_append")


def replay_extract(rb, env_idxs, step_idxs, obs, next_obs, actions, returns, terminals, stream=None):
    _want(env_idxs, torch.int32, "env_idxs")
    _want(step_idxs, torch.int32, "step_idxs")
    _check(load().arl_replay_extract(C.byref(rb), ptr(env_idxs), ptr(step_idxs), env_idxs.numel(), ptr(obs),
                                     ptr(next_obs), ptr(actions), ptr(returns), ptr(terminals),
                                     stream_ptr(stream)), "arl_replay_extract")


def sumtree_find(tree, levels, uniforms, out, stream=None):
    _want(tree, torch.float64, "tree")
    _want(uniforms, torch.float64, "uniforms")
    _want(out, torch.int32, "out")
    _check(load().arl_sumtree_find(ptr(tree), levels, ptr(uniforms), uniforms.numel(), ptr(out),
                                   stream_ptr(stream)), "arl_sumtree_find")


def sumtree_add(tree, levels, idxs, diffs, stream=None):
    _want(tree, torch.float64, "tree")
    _want(idxs, torch.int32, "idxs")
    _want(diffs, torch.float64, "diffs")
    _check(load().arl_sumtree_add(ptr(tree), levels, ptr(idxs), ptr(diffs), idxs.numel(), stream_ptr(stream)),
           "arl_sumtree_add")


def sumtree_sample(tree, levels, uniforms, n, part_size, tree_idxs, env_idxs, step_idxs, probs, n_unique,
                   stream=None):
    _want(tree, torch.float64, "tree"), _want(uniforms, torch.float64, "uniforms")
    _want(tree_idxs, torch.int32, "tree_idxs"), _want(probs, torch.float64, "probs")
    _check(load().arl_sumtree_sample(ptr(tree), levels, ptr(uniforms), uniforms.numel(), n, part_size,
                                     ptr(tree_idxs), ptr(env_idxs), ptr(step_idxs), ptr(probs), ptr(n_unique),
                                     stream_ptr(stream)), "arl_sumtree_sample")


def sumtree_sample_batch(tree, levels, uniforms, n, part_size, tree_idxs, env_idxs, step_idxs, probs, n_unique,
                         beta=0.0, is_weights=None, notify=None, ticket=0, stream=None):
    """sumtree_sample + the batch's importance weights in one launch; uniforms: device or PINNED host f64; notify: a pinned
    int64[1] that receives (ticket << 32) | n_unique when the launch is through (the host polls it)."""
    _want(tree, torch.float64, "tree"), _want(uniforms, torch.float64, "uniforms")
    _want(tree_idxs, torch.int32, "tree_idxs"), _want(probs, torch.float64, "probs")
    if is_weights is not None:
        _want(is_weights, torch.float32, "is_weights")
        assert is_weights.numel() == n
    if notify is not None:
        _want(notify, torch.int64, "notify")
        assert notify.is_pinned()
    _check(load().arl_sumtree_sample_batch(ptr(tree), levels, staged_ptr(uniforms), uniforms.numel(), n, part_size,
                                           ptr(tree_idxs), ptr(env_idxs), ptr(step_idxs), ptr(probs), ptr(n_unique),
                                           float(beta), ptr(is_weights), None if notify is None else notify.data_ptr(),
                                           int(ticket), stream_ptr(stream)), "arl_sumtree_sample_batch")


def sumtree_update_pow(tree, levels, idxs, priorities, last_probs, alpha, stream=None):
    """tree[path of idxs] += f32(priorities ** alpha) - last_probs, one launch (priority_diffs + sumtree_add)."""
    _want(tree, torch.float64, "tree"), _want(idxs, torch.int32, "idxs")
    _want(priorities, torch.float32, "priorities"), _want(last_probs, torch.float64, "last_probs")
    assert priorities.numel() == idxs.numel() == last_probs.numel()
    _check(load().arl_sumtree_update_pow(ptr(tree), levels, ptr(idxs), ptr(priorities), ptr(last_probs), float(alpha),
                                         idxs.numel(), stream_ptr(stream)), "arl_sumtree_update_pow")


def is_weights(probs, beta, out, stream=None):
    _want(probs, torch.float64, "probs"), _want(out, torch.float32, "out")
    _check(load().arl_is_weights(ptr(probs), probs.numel(), float(beta), ptr(out), stream_ptr(stream)),
           "arl_is_weights")


def priority_diffs(priorities, last_probs, alpha, diffs, stream=None):
    _want(priorities, torch.float32, "priorities"), _want(last_probs, torch.float64, "last_probs")
    _check(load().arl_priority_diffs(ptr(priorities), ptr(last_probs), priorities.numel(), float(alpha),
                                     ptr(diffs), stream_ptr(stream)), "arl_priority_diffs")


def sumtree_gather(tree, idxs, out, scale=1.0, stream=None):
    _want(tree, torch.float64, "tree")
    _want(idxs, torch.int32, "idxs")
    _want(out, torch.float64, "out")
    _check(load().arl_sumtree_gather(ptr(tree), ptr(idxs), idxs.numel(), float(scale), ptr(out),
                                     stream_ptr(stream)), "arl_sumtree_gather")


# ---------------------------------------------------------------------------
# categorical DQN output stage (csrc/dqn.hip)
# ---------------------------------------------------------------------------

def catdqn_act(logits, z, override, n_actions, n_atoms, onehot, greedy=None, dueling=False, stream=None):
    batch = onehot.shape[0]
    stride = logits.numel() // (batch * (n_actions + int(dueling)))
    _check(load().arl_catdqn_act(ptr(logits), ptr(z), ptr(override), batch, n_actions, n_atoms, stride,
                                 int(dueling), ptr(onehot), ptr(greedy), stream_ptr(stream)), "arl_catdqn_act")


def catdqn_loss(pred_logits, tgt_next_logits, pol_next_logits, z, actions, returns, terminals, is_weights,
                n_actions, n_atoms, v_min, v_max, gamma_n, dlogits, loss_rows, kl, dueling=False, stream=None):
    batch = actions.numel()
    stride = pred_logits.numel() // (batch * (n_actions + int(dueling)))
    _check(load().arl_catdqn_loss(ptr(pred_logits), ptr(tgt_next_logits), ptr(pol_next_logits), ptr(z),
                                  ptr(actions), ptr(returns), ptr(terminals), ptr(is_weights), batch, n_actions,
                                  n_atoms, stride, int(dueling), float(v_min), float(v_max), float(gamma_n),
                                  ptr(dlogits), ptr(loss_rows), ptr(kl), stream_ptr(stream)), "arl_catdqn_loss")


def logit_src(item, bias, row0=0, row_floats=0):
    """ArlLogitSrc for arl_catdqn_loss_parts from what conv2d_fwd_parts returned for the output layer: rows from `row0`
    on (row_floats floats per row); an unsplit launch (item.splits == 0: `out` is final) reads as one split, no bias."""
    src = ArlLogitSrc()
    split = item.splits > 0
    src.part = (item.part if split else item.out) + 4 * row0 * row_floats
    src.bias_or_null = bias.data_ptr() if (split and bias is not None) else None
    src.split_stride = item.total
    src.splits = item.splits if split else 1
    return src


def catdqn_loss_parts(pred, tgt_next, pol_next, z, actions, returns, terminals, is_weights, n_actions, n_atoms,
                      atom_stride, v_min, v_max, gamma_n, dlogits, loss_rows, kl, dueling=False, stream=None,
                      dgrad_weights=None):
    """arl_catdqn_loss on logit blocks still in split partial sums (ArlLogitSrc each; pol_next None: not double DQN).
    dgrad_weights: [(w, wt, geom)] -- the launch also writes these layers' k-contiguous weight copies."""
    wt_items, n_wt = None, 0
    if dgrad_weights:
        n_wt = len(dgrad_weights)
        assert n_wt <= DGRAD_WT_MAX
        wt_items = (ArlDgradWt * n_wt)()
        for it, (w, wt, geom) in zip(wt_items, dgrad_weights):
            it.w, it.wt, it.geom = ptr(w), ptr(wt), C.pointer(geom)
    _check(load().arl_catdqn_loss_parts(C.byref(pred), C.byref(tgt_next), None if pol_next is None else C.byref(pol_next),
                                        ptr(z), ptr(actions), ptr(returns), ptr(terminals), ptr(is_weights),
                                        actions.numel(), n_actions, n_atoms, atom_stride, int(dueling), float(v_min),
                                        float(v_max), float(gamma_n), ptr(dlogits), ptr(loss_rows), ptr(kl),
                                        None if wt_items is None else C.cast(wt_items, _vp), n_wt,
                                        stream_ptr(stream)), "arl_catdqn_loss_parts")


def dqn_act(q, override, n_actions, onehot, greedy=None, dueling=False, stream=None):
    batch = onehot.shape[0]
    _check(load().arl_dqn_act(ptr(q), ptr(override), batch, n_actions, q.numel() // batch, int(dueling),
                              ptr(onehot), ptr(greedy), stream_ptr(stream)), "arl_dqn_act")


def dqn_loss(q, tgt_next_q, pol_next_q, actions, returns, terminals, is_weights, n_actions, gamma_n, delta_clip,
             dq, loss_rows, td_abs, dueling=False, stream=None):
    """delta_clip None: squared loss."""
    batch = actions.numel()
    _check(load().arl_dqn_loss(ptr(q), ptr(tgt_next_q), ptr(pol_next_q), ptr(actions), ptr(returns),
                               ptr(terminals), ptr(is_weights), batch, n_actions, q.numel() // batch, int(dueling),
                               float(gamma_n), 0.0 if delta_clip is None else float(delta_clip), ptr(dq),
                               ptr(loss_rows), ptr(td_abs), stream_ptr(stream)), "arl_dqn_loss")


# ---------------------------------------------------------------------------
# LSTM cell (csrc/lstm.hip).  Tensors may be row-strided views ([B, cols] with stride(1) == 1).
# ---------------------------------------------------------------------------

def _rows(t):
    """(data_ptr, row stride in elements) of a 2-D tensor whose rows are contiguous."""
    if t is None:
        return None, 0
    assert t.dim() == 2 and t.stride(1) == 1 and t.is_cuda and t.dtype == torch.float32
    return t.data_ptr(), t.stride(0)


def lstm_cell_fwd(gx, gh, c_prev, h_out, c_out, gates=None, stream=None):
    batch, hidden = c_prev.shape
    (pgx, sgx), (pcp, scp), (ph, sh), (pc, sc), (pg, sg) = _rows(gx), _rows(c_prev), _rows(h_out), _rows(c_out), _rows(gates)
    _check(load().arl_lstm_cell_fwd(pgx, sgx, ptr(gh), pcp, scp, batch, hidden, ph, sh, pc, sc, pg, sg,
                                    stream_ptr(stream)), "arl_lstm_cell_fwd")


def lstm_cell_bwd(dh, dh_rec, dc_next, gates, c_prev, c_out, dgates, dc_prev, stream=None):
    batch, hidden = c_prev.shape
    (pdh, sdh), (pg, sg), (pcp, scp), (pc, sc), (pdg, sdg) = _rows(dh), _rows(gates), _rows(c_prev), _rows(c_out), _rows(dgates)
    _check(load().arl_lstm_cell_bwd(pdh, sdh, ptr(dh_rec), ptr(dc_next), pg, sg, pcp, scp, pc, sc, batch, hidden,
                                    pdg, sdg, ptr(dc_prev), stream_ptr(stream)), "arl_lstm_cell_bwd")


def gru_cell_fwd(gx, gh, h_prev, h_out, saved=None, stream=None):
    batch, hidden = h_prev.shape
    (pgx, sgx), (php, shp), (ph, sh), (ps, ss) = _rows(gx), _rows(h_prev), _rows(h_out), _rows(saved)
    _check(load().arl_gru_cell_fwd(pgx, sgx, ptr(gh), php, shp, batch, hidden, ph, sh, ps, ss, stream_ptr(stream)),
           "arl_gru_cell_fwd")


def gru_cell_bwd(dh, dh_rec, dh_dir, saved, h_prev, dgx, dgh, dh_prev, stream=None):
    batch, hidden = h_prev.shape
    (pdh, sdh), (ps, ss), (php, shp), (pgx, sgx), (pgh, sgh) = _rows(dh), _rows(saved), _rows(h_prev), _rows(dgx), _rows(dgh)
    _check(load().arl_gru_cell_bwd(pdh, sdh, ptr(dh_rec), ptr(dh_dir), ps, ss, php, shp, batch, hidden, pgx, sgx,
                                   pgh, sgh, ptr(dh_prev), stream_ptr(stream)), "arl_gru_cell_bwd")


def rnn_cell_fwd(gx, gh, h_out, stream=None):
    batch, hidden = h_out.shape
    (pgx, sgx), (ph, sh) = _rows(gx), _rows(h_out)
    _check(load().arl_rnn_cell_fwd(pgx, sgx, ptr(gh), batch, hidden, ph, sh, stream_ptr(stream)), "arl_rnn_cell_fwd")


def rnn_cell_bwd(dh, dh_rec, h_out, dpre, stream=None):
    batch, hidden = h_out.shape
    (pdh, sdh), (ph, sh), (pd, sd) = _rows(dh), _rows(h_out), _rows(dpre)
    _check(load().arl_rnn_cell_bwd(pdh, sdh, ptr(dh_rec), ph, sh, batch, hidden, pd, sd, stream_ptr(stream)),
           "arl_rnn_cell_bwd")
