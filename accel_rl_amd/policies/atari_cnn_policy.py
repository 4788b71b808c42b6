"""Feed-forward Atari CNN policy: conv stack -> dense -> (softmax pi, linear V).

Mirror of the reference's AtariCnnPolicy / PgCnn
(accel_rl/policies/pg/atari_cnn_policy.py:15-119, pg/networks/pg_cnn.py:11-119,
policies/layers.py:11-41).  Every stage is hand-written HIP: the dense contractions
(the only MFMA-shaped stage of the path) are the fp32-MFMA implicit-GEMM kernels of
csrc/mfma_conv.hip -- deterministic, no atomics, so seeded runs reproduce bit for bit
-- and everything between them is csrc/learner.hip, all on channels-last activations:

  forward :  conv 1 + bias + relu straight from the u8 observations (rows picked by the minibatch
             indices and scaled by 1/255 inside the kernel's loader; other first layers:
             gather+scale u8 NCHW -> f32 NHWC first)  ->  [conv + bias + relu] x (n - 1)
             -> dense + bias + relu -> heads+softmax (infer) | heads+losses+grads (train)
  backward:  head wgrad -> [relu-bwd+bias-grad -> dense dW / dx] -> [relu-bwd+bias-grad ->
             conv dW / dx] x n, every gradient written straight into ONE flat
             fp32 bucket (`flat_grads`) that the HIP optimiser / RCCL all-reduce consume.

No autograd graph is ever built and no MIOpen / hipBLASLt kernel runs: the HIP kernels are the only
backend (the autograd formulation of the same network used by the numerics tests lives in
tests/autograd_ref.py).

Internal parameter layout (the bucket is ours to define): conv W as (K,kh,kw,C)
= channels-last correlation kernels (conv 1 on the u8 path: (K,C,kh,kw), its input being
planar); first dense W with its input columns in
(h,w,c) order; both heads fused in one matrix W_head[(A+1), hid].
get/set_param_values convert to/from the reference's flat vector: order conv_i.W,
conv_i.b, hidden_i.W, hidden_i.b, output_pi.W, .b, output_v.W, .b
(pg_cnn.py:47-86,118-119), conv W (out,in,kh,kw) for a flipped (true) convolution,
dense W (in,out) with (c,h,w) input order.
"""
import ctypes as C
import os

import numpy as np
import torch

from accel_rl_amd import _lib
from accel_rl_amd.distributions import Categorical
from accel_rl_amd.spaces import Discrete
from accel_rl_amd.util.seed import layer_rng

KIND_A2C, KIND_PPO = 0, 1


def _glorot_uniform(shape):
    """Lasagne GlorotUniform for a conv filter bank (out, in, kh, kw)."""
    receptive = int(np.prod(shape[2:]))
    fan_in, fan_out = shape[1] * receptive, shape[0] * receptive
    a = np.sqrt(6. / (fan_in + fan_out))
    return layer_rng().uniform(low=-a, high=a, size=shape).astype(np.float32)


def _norm_c(shape, std):
    """NormCInit (reference: policies/layers.py:11-19); shape = (in, out)."""
    out = np.random.randn(*shape).astype(np.float32)
    out *= std / np.sqrt(np.square(out).sum(axis=0, keepdims=True))
    return out


class ObsRows(object):
    """Rows of a u8 observation array that the first convolution reads in place: obs[idx] (all rows
    without idx), scaled inside the kernel.  Stands where the scaled f32 input tensor would."""
    __slots__ = ("obs", "idx", "shape")

    def __init__(self, obs, idx):
        self.obs, self.idx = obs.contiguous(), idx
        self.shape = (obs.shape[0] if idx is None else idx.shape[0],) + tuple(obs.shape[1:])


class AtariCnnPolicy(object):

    # conv 1 reads the u8 observations directly when its geometry allows (subclasses that build their
    # own scaled inputs switch this off)
    _u8_conv1 = True
    # prob_value(observations, rows) reads rows of a larger buffer in place (GpuVecSampler then keeps no second,
    # contiguous copy of the current observations up to date); subclasses with their own prob_value say no
    serves_rows = True

    def __init__(self, conv_filters, conv_filter_sizes, conv_strides, conv_pads,
                 hidden_sizes=(), pixel_scale=255., initial_param_values=None):
        self.conv_filters, self.conv_filter_sizes = list(conv_filters), list(conv_filter_sizes)
        self.conv_strides, self.conv_pads = list(conv_strides), list(conv_pads)
        self.hidden_sizes = list(hidden_sizes)
        if not self.hidden_sizes:
            raise NotImplementedError("at least one hidden dense layer is required")
        self.pixel_scale = pixel_scale
        self.initial_param_values = initial_param_values
        self._scratch = dict()

    recurrent = property(lambda self: False)
    vectorized = property(lambda self: True)
    state_info_keys = property(lambda self: [])
    distribution = property(lambda self: self._dist)

    # ---------------------------------------------------------------- build
    def initialize(self, env_spec, device=None, **kwargs):
        assert isinstance(env_spec.action_space, Discrete)
        _lib.load()
        self.device = torch.device("cuda:0" if device is None else device)
        self.env_spec = env_spec
        self.action_space = env_spec.action_space
        c, h, w = env_spec.observation_space.shape
        n_act = self.n_act = self.action_space.n
        self._obs_shape = (c, h, w)
        self._c_in, self._c_pad = c, (c + 3) // 4 * 4
        # ---- reference-layout initial values, drawn in the reference's order
        ref, self._conv_geom = [], []
        nf0, sz0, st0, pad0 = self.conv_filters[0], self.conv_filter_sizes[0], self.conv_strides[0], self.conv_pads[0]
        self._u8 = bool(self._u8_conv1 and
                        _lib.conv2d_u8_supported(h, w, nf0, sz0, sz0, st0, pad0[0], pad0[1]))
        for nf, sz, st, pad in zip(self.conv_filters, self.conv_filter_sizes, self.conv_strides,
                                   self.conv_pads):
            ref += [_glorot_uniform((nf, c, sz, sz)), np.zeros(nf, np.float32)]
            h = (h + 2 * pad[0] - sz) // st + 1
            w = (w + 2 * pad[1] - sz) // st + 1
            if nf % 4:
                raise NotImplementedError("conv filter counts must be multiples of 4 (got %d)" % nf)
            # the MFMA kernels want channel counts in multiples of 4: a 1-, 2- or 3-frame stack is
            # zero-padded to 4 input channels internally (the extra weights stay exactly zero)
            # (the u8 path takes any plane count)
            ci = c if (self._u8 and not self._conv_geom) else (c + 3) // 4 * 4
            self._conv_geom.append((nf, ci, sz, st, tuple(pad), h, w))
            c = nf
        self._conv_out = (c, h, w)
        fan = c * h * w
        hid_ref, hid_names, fan = self._hidden_reference_init(fan)
        self._n_hidden_tensors = len(hid_ref)
        head_ref, head_names = self._head_reference_init(fan, n_act)
        ref += self._tail_to_flat(hid_ref + head_ref)
        self._ref_shapes = [a.shape for a in ref]
        self.param_short_names = (["Conv%d%s" % (i, s) for i in range(len(self._conv_geom)) for s in "Wb"] +
                                  self._tail_to_flat(hid_names + head_names))
        self.n_params = int(sum(a.size for a in ref))
        # ---- internal bucket: [conv W, b]... [hidden W, b]... W_head, b_head
        shapes = []
        for i, (nf, ci, sz, st, pad, ho, wo) in enumerate(self._conv_geom):
            shapes += [(nf, ci, sz, sz) if (i == 0 and self._u8) else (nf, sz, sz, ci), (nf,)]
        hid_shapes = self._hidden_internal_shapes()
        self._n_hidden_internal = len(hid_shapes)       # may differ from the reference's count (GRU: 6 vs 12)
        shapes += hid_shapes
        shapes += self._head_internal_shapes(fan, n_act)
        self._k_head = 2 * len(self._conv_geom) + self._n_hidden_internal    # index of W_head in params / grads
        self._shapes = shapes
        sizes = [int(np.prod(s)) for s in shapes]
        self._offsets, off = [], 0
        for n in sizes:                       # every tensor starts on a 16-byte boundary
            self._offsets.append(off)
            off += (n + 3) // 4 * 4
        self._bucket_len = off
        # everything from the first dense tensor on (dense layers, heads) is final once the dense layers'
        # backward has run: the sync optimizers all-reduce that tail under the conv layers' backward
        self.grad_split_offset = self._offsets[2 * len(self._conv_geom)]
        self.flat_params = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.flat_grads = torch.zeros(off, dtype=torch.float32, device=self.device)

        def views(flat):
            out = []
            for k, (o, n, s) in enumerate(zip(self._offsets, sizes, shapes)):
                v = flat[o:o + n].view(s)
                out.append(v.permute(0, 3, 1, 2) if (len(s) == 4 and not (k == 0 and self._u8)) else v)   # logical (K,C,kh,kw)
            return out
        self.params = [torch.nn.Parameter(v) for v in views(self.flat_params)]
        self.grads = views(self.flat_grads)
        # raw (memory-order) views for the HIP kernels: conv W is (K, kh, kw, C) in memory ((K, C, kh, kw): u8 conv 1)
        self._w = [self.flat_params[o:o + n] for o, n in zip(self._offsets, sizes)]
        self._g = [self.flat_grads[o:o + n] for o, n in zip(self._offsets, sizes)]
        self._conv_ws = _lib.conv_workspace(self.device)
        # the data gradients' own copy of the conv weights (reduction index contiguous: _lib.conv2d_dgrad_weights), for
        # the layers whose data gradient runs on the split kernels of 17 .. 64 columns; refreshed once per backward pass
        self._wt = {i: torch.empty_like(self._w[2 * i]) for i, (nf, ci, sz, st, pad, ho, wo) in enumerate(self._conv_geom)
                    if i > 0 and 16 < ci <= 64 and st <= 2 and sz % st == 0 and
                    os.environ.get("ARL_DGRAD_WT", "1") != "0"}                     # (A/B switch)
        self._geoms = dict()
        for p, g in zip(self.params, self.grads):
            p.grad = g
        self._n_conv, self._n_hid = len(self._conv_geom), len(self._hid_geom)
        self._dist = Categorical(n_act)
        self._scale = float(np.float32(1. / self.pixel_scale))
        self._relu_ws = _lib.relu_bwd_workspace(self.device)
        self._folds, self._fold_workspaces = _lib.FoldList(), {}
        self._loss_ws = _lib.pg_head_workspace(self.device)
        self._set_from_reference_arrays(ref)
        if self.initial_param_values is not None:
            self.set_param_values(self.initial_param_values)

    # ---- hidden layers between the conv stack and the output layers: dense + relu
    #      (pg_cnn.py:57-68); recurrent subclasses override these four
    def _hidden_reference_init(self, fan):
        ref, names, self._hid_geom = [], [], []
        for i, hs in enumerate(self.hidden_sizes):
            if hs % 4:
                raise NotImplementedError("hidden sizes must be multiples of 4 (got %d)" % hs)
            ref += [_norm_c((fan, hs), 1.0), np.zeros(hs, np.float32)]
            names += ["FC%dW" % i, "FC%db" % i]
            self._hid_geom.append((hs, fan))
            fan = hs
        return ref, names, fan

    def _hidden_internal_shapes(self):
        return [s for hs, fan_in in self._hid_geom for s in ((hs, fan_in), (hs,))]

    def _conv_flat_to_internal(self, w_ref):
        """(c*h*w, units) reference dense weight on the conv output -> (units, h*w*c) internal."""
        co, ho, wo = self._conv_out
        units = w_ref.shape[1]
        return w_ref.reshape(co, ho, wo, units).transpose(3, 1, 2, 0).reshape(units, -1)

    def _conv_flat_to_reference(self, w_int):
        co, ho, wo = self._conv_out
        units = w_int.shape[0]
        return w_int.reshape(units, ho, wo, co).transpose(3, 1, 2, 0).reshape(-1, units)

    def _hidden_to_reference(self, arrs):
        out = []
        for j in range(len(self._hid_geom)):
            w = arrs[2 * j]
            out += [self._conv_flat_to_reference(w) if j == 0 else w.T, arrs[2 * j + 1]]
        return out

    def _hidden_to_internal(self, refs):
        out = []
        for j in range(len(self._hid_geom)):
            w = refs[2 * j]
            out += [self._conv_flat_to_internal(w) if j == 0 else w.T, refs[2 * j + 1]]
        return out

    # The hooks above and below produce / consume the dense tensors in CONSTRUCTION order (hidden layers, then
    # output layers).  Where the reference's flat parameter vector orders them differently (dueling networks:
    # get_all_params walks the value branch first), `_tail_perm` lists, per flat position, the index into
    # the construction-order list.
    _tail_perm = None

    def _tail_to_flat(self, tail):
        return list(tail) if self._tail_perm is None else [tail[i] for i in self._tail_perm]

    def _tail_from_flat(self, tail):
        if self._tail_perm is None:
            return list(tail)
        out = [None] * len(tail)
        for pos, i in enumerate(self._tail_perm):
            out[i] = tail[pos]
        return out

    # ---- output layers: policy + value heads fused in one matrix W_head[(A+1), hid]
    #      (pg_cnn.py:70-86); subclasses with other output layers override these four
    def _head_reference_init(self, fan, n_act):
        return ([_norm_c((fan, n_act), 0.01), np.zeros(n_act, np.float32),
                 _norm_c((fan, 1), 1.0), np.zeros(1, np.float32)], ["OutputW", "Outputb", "OutputW", "Outputb"])

    def _head_internal_shapes(self, fan, n_act):
        return [(n_act + 1, fan), (n_act + 1,)]

    def _head_to_reference(self, wh, bh):
        a = self.n_act
        return [wh[:a].T, bh[:a], wh[a:].T, bh[a:]]

    def _head_to_internal(self, ref_tail):
        return [np.concatenate([ref_tail[0].T, ref_tail[2].T], axis=0), np.concatenate([ref_tail[1], ref_tail[3]])]

    # -------------------------------------------------------------- forward
    def _buffer(self, key, shape, channels_last=False, dtype=torch.float32):
        """Static scratch tensor (re-used across calls; fresh inside a graph capture)."""
        capturing = torch.cuda.is_current_stream_capturing()
        buf = None if capturing else self._scratch.get(key)
        if buf is None:
            buf = torch.empty(shape, dtype=dtype, device=self.device,
                              memory_format=torch.channels_last if channels_last
                              else torch.contiguous_format)
            if not capturing:
                self._scratch[key] = buf
        return buf

    def _scaled(self, obs_u8, idx=None, tag=""):
        """The network input for u8 [n,C,H,W] observations (rows optionally gathered by idx): on the u8
        path just the (obs, idx) pair -- conv 1 scales while it loads --, else `_scaled_f32`."""
        if self._u8:
            return ObsRows(obs_u8, idx)
        return self._scaled_f32(obs_u8, idx, tag)

    def _scaled_f32(self, obs_u8, idx=None, tag=""):
        """u8 [n,C,H,W] (rows optionally gathered by idx) -> f32 * (1/pixel_scale),
        logical [B,C,H,W] in channels-last memory."""
        b = obs_u8.shape[0] if idx is None else idx.shape[0]
        c, h, w = self._obs_shape
        if c == 4:
            out = self._buffer(("x" + tag, b), (b, c, h, w), channels_last=True)
            _lib.gather_scale_obs_nhwc(obs_u8, idx, out, self._scale)
            return out
        tmp = self._buffer(("x_nchw" + tag, b), (b, c, h, w))
        _lib.gather_scale_obs(obs_u8, idx, tmp, self._scale)
        out = self._buffer(("x" + tag, b), (b, self._c_pad, h, w), channels_last=True)
        if self._c_pad != c:
            out.zero_()
        out[:, :c].copy_(tmp)
        return out

    def _layer_geoms(self, b):
        """ctypes geometry records of every layer at batch size b (cached per contraction route: _lib.default_route)."""
        key = (b, _lib.default_route)
        gs = self._geoms.get(key)
        if gs is None:
            c, h, w = self._obs_shape
            conv = []
            for nf, ci, sz, st, pad, ho, wo in self._conv_geom:
                conv.append(_lib.conv_geom(b, h, w, ci, nf, sz, sz, st, pad[0], pad[1]))
                h, w = ho, wo
            dense = [_lib.dense_geom(b, fan_in, hs) for hs, fan_in in self._hid_geom]
            gs = self._geoms[key] = (conv, dense)
        return gs

    def _trunk(self, x, w=None, tag=""):
        """Explicit conv/dense stack (no autograd) on NHWC memory.  Returns (conv activations
        [B,Ho,Wo,K], hidden activations [B,units]); every activation is post bias+relu.
        `w`: the layers' (W, b) views to use (default: the trainable ones); `tag` keeps the scratch
        activations of a second forward (e.g. a target network) apart."""
        b = x.shape[0]
        w = self._w if w is None else w
        conv_g, dense_g = self._layer_geoms(b)
        acts, a = [], x
        for i, (nf, ci, sz, st, pad, ho, wo) in enumerate(self._conv_geom):
            z = self._buffer(("act" + tag, i, b), (b, ho, wo, nf))
            if isinstance(a, ObsRows):
                _lib.conv2d_u8_fwd(a.obs, a.idx, self._scale, w[0], w[1], z, conv_g[0], True)
            else:
                _lib.conv2d_fwd(a, w[2 * i], w[2 * i + 1], z, conv_g[i], True, self._conv_ws)
            acts.append(z)
            a = z
        hids, k = [], 2 * self._n_conv
        for j, (hs, fan_in) in enumerate(self._hid_geom):
            hcur = self._buffer(("hid" + tag, j, b), (b, hs))
            _lib.conv2d_fwd(a, w[k], w[k + 1], hcur, dense_g[j], True, self._conv_ws)
            hids.append(hcur)
            a = hcur
            k += 2
        return acts, hids

    def prob_value(self, observations, rows=None):
        """Batched inference on device uint8 observations (the sampler's hot call;
        reference: _f_prob_value, atari_cnn_policy.py:67,109).  rows (i32[B]): serve observations[rows]
        without materialising them (the sampler's current observations are rows of its rollout buffer)."""
        with torch.no_grad():
            x = self._scaled(observations, rows)
            b = x.shape[0]
            _, hids = self._trunk(x)
            prob = torch.empty((b, self.n_act), dtype=torch.float32, device=self.device)
            value = torch.empty(b, dtype=torch.float32, device=self.device)
            _lib.pg_head_infer(hids[-1], self.params[-2], self.params[-1], prob, value)
            return prob, value

    # ------------------------------------------------- serving inside the env step's launch
    def serve_supported(self):
        """Can GpuVecSampler hand this policy's output layers to arl_env_step_served?  Only the plain feed-forward
        network (subclasses with their own prob_value -- recurrent, Q-networks -- keep the separate launches)."""
        return (type(self).prob_value is AtariCnnPolicy.prob_value and self._n_hid >= 1 and
                self._hid_geom[-1][0] <= 1024 and self.n_act <= _lib.MAX_ACTIONS)

    def serve_conv1(self, game, b):
        """(_lib.ArlServeConv1, y1) for a launch that computes conv 1 of b observations itself (arl_env_step_served,
        arl_rollout_begin_conv1), or (None, None) when this first layer / route is not one it takes."""
        conv_g, _ = self._layer_geoms(b)
        if not (self._u8 and _lib.serve_conv1_supported(game, conv_g[0])):
            return None, None
        nf, ci, sz, st, pad, ho, wo = self._conv_geom[0]
        y1 = self._buffer(("act", 0, b), (b, ho, wo, nf))
        conv1 = _lib.ArlServeConv1()
        conv1.geom = C.pointer(conv_g[0])
        conv1.w, conv1.bias, conv1.y = self._w[0].data_ptr(), self._w[1].data_ptr(), y1.data_ptr()
        conv1.scale, conv1.relu = self._scale, 1
        return conv1, y1

    def serve_forward(self, game, observations, rows, y1=None, want_next=True):
        """prob_value's trunk for arl_env_step_served: everything up to the last hidden layer, whose split partial sums
        stay unfolded (the step launch folds them, applies bias + rectifier, evaluates the heads and samples).
        y1: conv 1's output for these rows if the previous step's launch has already computed it (None: computed here).
        Returns (head, conv1, y1_next): _lib.ArlServeHead; _lib.ArlServeConv1 or None (want_next False, or a first layer
        the step launch does not take); the tensor that launch will leave the next rows' conv 1 output in (or None).
        Same kernels, same arithmetic as prob_value (atari_cnn_policy.py:63-67 of the reference)."""
        with torch.no_grad():
            b = rows.shape[0]
            conv_g, dense_g = self._layer_geoms(b)
            w = self._w
            nf, ci, sz, st, pad, ho, wo = self._conv_geom[0]
            if y1 is None:
                x = self._scaled(observations, rows)
                y1 = self._buffer(("act", 0, b), (b, ho, wo, nf))
                if isinstance(x, ObsRows):
                    _lib.conv2d_u8_fwd(x.obs, x.idx, self._scale, w[0], w[1], y1, conv_g[0], True)
                else:
                    _lib.conv2d_fwd(x, w[0], w[1], y1, conv_g[0], True, self._conv_ws)
            a = y1
            for i in range(1, self._n_conv):
                nf_i, _, _, _, _, ho_i, wo_i = self._conv_geom[i]
                z = self._buffer(("act", i, b), (b, ho_i, wo_i, nf_i))
                _lib.conv2d_fwd(a, w[2 * i], w[2 * i + 1], z, conv_g[i], True, self._conv_ws)
                a = z
            k = 2 * self._n_conv
            for j, (hs, fan_in) in enumerate(self._hid_geom):
                hcur = self._buffer(("hid", j, b), (b, hs))
                if j + 1 < self._n_hid:
                    _lib.conv2d_fwd(a, w[k], w[k + 1], hcur, dense_g[j], True, self._conv_ws)
                else:
                    item = _lib.conv2d_fwd_parts(a, w[k], w[k + 1], hcur, dense_g[j], True, self._conv_ws)
                a = hcur
                k += 2
            head = _lib.ArlServeHead()
            head.hidden = item
            head.hidden_bias, head.hidden_relu, head.hid = w[k - 1].data_ptr(), 1, self._hid_geom[-1][0]
            head.w_head, head.b_head = w[self._k_head].data_ptr(), w[self._k_head + 1].data_ptr()
            conv1 = None
            if want_next and self._u8 and _lib.serve_conv1_supported(game, conv_g[0]):
                conv1 = _lib.ArlServeConv1()            # (into the buffer this step's conv 2 has just read: stream order)
                conv1.geom = C.pointer(conv_g[0])
                conv1.w, conv1.bias, conv1.y = w[0].data_ptr(), w[1].data_ptr(), y1.data_ptr()
                conv1.scale, conv1.relu = self._scale, 1
            return head, conv1, (y1 if conv1 is not None else None)

    # ------------------------------------------------------------- training
    def loss_and_grads(self, mb, kind, clip_param, v_loss_coeff, ent_loss_coeff, lr_mult,
                       inv_count=None, tie_rule=_lib.PPO_TIE_THEANO):
        """One minibatch: forward, losses (a2c.py:43-46 / ppo.py:42-51 + aac_base.py:60-66)
        and the full backward pass into `flat_grads` (overwritten, not accumulated).
        mb: observations u8[n,...], idx i32[B] or None, actions, advantages, returns,
        old_prob, valids (full-batch arrays, rows selected by idx).  Returns loss4 =
        (pi_loss, v_loss, ent_loss, pi+v+ent) as a device tensor.  tie_rule (PPO): whose gradient the surrogate's
        min() / clip() hand on -- the reference's Theano graph (default) or the mathematical derivative
        (ARL_PPO_TIE_*, accel_rl_hip.h)."""
        with torch.no_grad():
            idx = mb.get("idx")
            rows = mb["observations"].shape[0] if idx is None else idx.shape[0]
            limit = self.rows_per_pass()
            if limit and rows >= 3 * limit:
                return self._loss_and_grads_in_passes(mb, rows, limit, kind, clip_param, v_loss_coeff, ent_loss_coeff,
                                                      lr_mult, inv_count, tie_rule)
            x = self._scaled(mb["observations"], idx)
            b = x.shape[0]
            acts, hids = self._trunk(x)
            g, k_head = self.grads, self._k_head
            hid = self._hid_geom[-1][0]
            dout = self._buffer(("dout", b), (b, self.n_act + 1))
            dh = self._buffer(("dh", b), (b, hid))
            loss4 = self._buffer(("loss", b), (4,))
            # (the head's launch also writes the data gradients' weight copies: _backward_convs then skips its own launch)
            wts = self._dgrad_weight_items(b) if os.environ.get("ARL_WT_IN_HEAD", "1") != "0" else []      # (A/B switch)
            self._folds.pg_head_loss(hids[-1], self.params[k_head], self.params[k_head + 1], mb["actions"],
                              mb["advantages"], mb["returns"], mb.get("old_prob"), mb.get("valids"),
                              idx, lr_mult, inv_count, self.n_act, kind, clip_param, v_loss_coeff,
                              ent_loss_coeff, dout, dh, g[k_head], g[k_head + 1], loss4, self._loss_ws,
                              relu_mask_dh=True, tie_rule=tie_rule, dgrad_weights=wts)
            self._wt_fresh = bool(wts)
            try:
                self._backward_trunk(x, acts, hids, dh, masked=True, split_hook=mb.get("split_hook"),
                                     dense_w_hook=mb.get("dense_w_hook"))
            finally:
                self._wt_fresh = False
            return loss4

    # Rows of a minibatch per forward + backward pass: None (the default) = one pass whatever the size; a number, or
    # "auto" (as many rows as keep a pass's activations, their gradients and the u8 rows it reads inside the 256 MiB
    # Infinity Cache: 1024 for spec 1, 2304 for spec 0), walks a minibatch of three or more such passes in passes whose
    # gradients are added in a fixed order -- the same mean-loss gradient, other summation order
    # (accel_rl/optimizers/single/a2c_optimizer.py:37-43 takes one step on the whole batch; so does this).  "auto" was the
    # default from round 3 (cost per row of spec 1 15-25 % higher at 4096-5120 rows in one pass,
    # profiles/r03/batch_sweep.txt) until round 6 found that rise in ONE launch -- the dense layer's data + weight
    # gradient sharing a launch, now bounded by size (arl_conv2d_bwd_pair) -- and not in the cache: with it gone one pass
    # is the cheaper walk at every size (spec 1 at 4096 rows: 2 782 us in one pass, 3 144 in passes of 1024;
    # profiles/r06/batch_sweep_passes.txt), so the option stays for memory-bound callers only.
    max_rows_per_pass = None
    CACHE_BYTES = 256 << 20

    def rows_per_pass(self):
        if not isinstance(self.max_rows_per_pass, str):
            return self.max_rows_per_pass
        per_row = int(np.prod(self._obs_shape)) + 8 * sum(nf * ho * wo for nf, ci, sz, st, pad, ho, wo in self._conv_geom)
        return max(256, (self.CACHE_BYTES // per_row + 128) // 256 * 256)

    def _loss_and_grads_in_passes(self, mb, rows, limit, kind, clip_param, v_loss_coeff, ent_loss_coeff, lr_mult,
                                  inv_count, tie_rule=_lib.PPO_TIE_THEANO):
        """loss_and_grads of a minibatch larger than rows_per_pass(): passes over consecutive slices of its index list,
        every pass normalised by the WHOLE minibatch's count; gradients and the four loss sums accumulate in pass
        order (deterministic).  The co-run / split hooks of the one-pass learner do not apply (nothing is final before
        the last pass)."""
        idx = mb.get("idx")
        if idx is None:
            idx = self._buffer(("all_rows", rows), (rows,), dtype=torch.int32)
            idx.copy_(torch.arange(rows, dtype=torch.int32, device=self.device))
        if inv_count is None:
            inv_count = self._buffer(("inv_rows", rows), (1,))
            inv_count.fill_(1.0 / rows)
        if not getattr(self, "_passes_logged", False):
            self._passes_logged = True
            from accel_rl_amd.util import logger
            logger.log("AtariCnnPolicy: minibatches of %d rows are walked in passes of %d (gradients accumulated in a fixed "
                       "order; max_rows_per_pass = None keeps one pass)" % (rows, limit))
        acc = self._buffer(("grad_acc",), tuple(self.flat_grads.shape))
        loss_acc = self._buffer(("loss_acc",), (4,))
        one = dict((k, v) for k, v in mb.items() if k not in ("split_hook", "dense_w_hook"))
        saved, self.max_rows_per_pass = self.max_rows_per_pass, None
        try:
            for k, lo in enumerate(range(0, rows, limit)):
                one["idx"] = idx[lo:min(lo + limit, rows)]
                loss4 = self.loss_and_grads(one, kind, clip_param, v_loss_coeff, ent_loss_coeff, lr_mult, inv_count,
                                            tie_rule)
                if k == 0:
                    acc.copy_(self.flat_grads)
                    loss_acc.copy_(loss4)
                else:
                    acc.add_(self.flat_grads)
                    loss_acc.add_(loss4)
        finally:
            self.max_rows_per_pass = saved
        self.flat_grads.copy_(acc)
        hook = mb.get("split_hook")
        if hook is not None:
            hook()                              # the whole bucket is final from here on
        return loss_acc

    def _backward_trunk(self, x, acts, hids, dh, masked=False, split_hook=None, dense_w_hook=None):
        """Gradients of every trunk layer into flat_grads, given dh = d loss / d (last hidden
        activation); masked: dh is already multiplied by that activation's rectifier mask.
        x, acts, hids as returned by _scaled / _trunk.
        dense_w_hook(first, count) -> job or None: called once the FIRST dense layer's weight gradient -- by far the
        largest tensor of the bucket, elements [first, first + count) -- is final in flat_grads (its kernel wrote it in
        place, no fold pending), i.e. before the conv layers' backward: the optimizer may answer with that range's update
        as a job (_lib.corun_job) which the next data-gradient launch that can carries in extra workgroups; if none
        does, it runs as its own launch at the end of the pass."""
        b = x.shape[0]
        conv_g, dense_g = self._layer_geoms(b)
        # ---- dense layers, last to first; the split folds of the whole pass run once, at the end
        d_cur, job = dh, None
        for j in range(self._n_hid - 1, -1, -1):
            k = 2 * (self._n_conv + j)
            hs, fan_in = self._hid_geom[j]
            inp = hids[j - 1] if j > 0 else acts[-1]
            d_prev = self._buffer(("dx_hid", j, b), (b, fan_in))
            self._layer_grads(d_cur, masked, hids[j], b, hs, k, dense_g[j], inp, d_prev)
            if j == 0 and dense_w_hook is not None and self._folds.last_dw_in_place:
                job = dense_w_hook(self._offsets[k], (self._g[k].numel() + 3) // 4 * 4)
            d_cur, masked = d_prev, True
        if split_hook is not None and self._n_hid:      # dense + head gradients are final from here on
            self._folds.run()
            split_hook()
        self._backward_convs(x, acts, d_cur, masked, corun=job)

    def _dgrad_weight_items(self, b):
        """[(w, wt, geom)] of the layers whose data gradient reads a k-contiguous copy of its weights (split routes)."""
        if not self._wt or _lib.default_route == _lib.ROUTE_FP32:
            return []
        conv_g, _ = self._layer_geoms(b)
        return [(self._w[2 * i], wt, conv_g[i]) for i, wt in sorted(self._wt.items())]

    def _backward_convs(self, x, acts, d_act, masked=False, corun=None):
        """Conv layers, last to first; d_act = NHWC gradient of the last conv output (masked: already
        multiplied by its rectifier mask); corun: an optimiser job (_lib.corun_job) for the first data-gradient launch
        that can carry it."""
        b = x.shape[0]
        conv_g, _ = self._layer_geoms(b)
        if not getattr(self, "_wt_fresh", False):          # (else: written by the head's launch of this pass)
            items = self._dgrad_weight_items(b)
            if items:
                _lib.conv2d_dgrad_weights(items)
        for i in range(self._n_conv - 1, -1, -1):
            nf, ci, sz, st, pad, ho, wo = self._conv_geom[i]
            d_in = self._buffer(("dx_conv", i, b), tuple(acts[i - 1].shape)) if i > 0 else None
            self._layer_grads(d_act, masked, acts[i], b * ho * wo, nf, 2 * i, conv_g[i], acts[i - 1] if i > 0 else x, d_in,
                              corun=corun if i > 0 else None,
                              wt=self._wt.get(i) if _lib.default_route != _lib.ROUTE_FP32 else None)
            if corun is not None and i > 0 and self._folds.corun_taken:
                corun = None
            d_act, masked = d_in, True
        if corun is not None:
            _lib.corun_job_run(corun)           # no launch could carry it: on its own, ahead of the step's update
        self._folds.run()

    def _layer_grads(self, d, masked, y, rows, channels, k, geom, inp, d_in, corun=None, wt=None):
        """One layer's backward: bias and weight gradient (deferred folds) and, with d_in, the data gradient
        already multiplied by the rectifier mask of `inp` (the layer below's output), so that the layer below
        gets its pre-activation gradient without another pass.  Not `masked`: d still needs this layer's own
        mask, applied in place by the streaming kernel that also sums the bias gradient."""
        folds, g = self._folds, self.grads
        if not masked:
            folds.relu_bwd_bias_grad(d, y, rows, channels, g[k + 1], self._fold_ws(("db", k)))
        dbias = g[k + 1] if masked else None
        if isinstance(inp, ObsRows):
            done = folds.conv2d_u8_bwd_weight(d, inp.obs, inp.idx, self._scale, self._g[k], geom,
                                              self._fold_ws(("dw", k)), dbias=dbias)
        elif d_in is not None:      # data + weight gradient share one launch where that pays (dense layers)
            done = folds.conv2d_bwd_pair(d, self._w[k], inp, d_in, inp, self._g[k], geom, self._fold_ws(("dw", k)),
                                         dbias=dbias, corun=corun, wt=wt)
        else:
            done = folds.conv2d_bwd_weight(d, inp, self._g[k], geom, self._fold_ws(("dw", k)), dbias=dbias)
        if not done:                # generic kernels leave the bias sums to the streaming kernel (its mask is idempotent)
            folds.relu_bwd_bias_grad(d, y, rows, channels, g[k + 1], self._fold_ws(("db", k)))

    def _fold_ws(self, key):
        """One workspace per pending fold (the partials stay live until folds.run())."""
        ws = self._fold_workspaces.get(key)
        if ws is None:
            ws = self._fold_workspaces[key] = (_lib.relu_bwd_workspace(self.device) if key[0] == "db"
                                               else _lib.conv_workspace(self.device))
        return ws

    def dist_info(self, observations, state_infos=None):
        return dict(prob=self.prob_value(observations)[0])

    def value(self, observations, state_infos=None):
        return self.prob_value(observations)[1]

    def dist_info_value(self, observations, state_infos=None):
        prob, value = self.prob_value(observations)
        return dict(prob=prob, value=value)

    # --------------------------------------------------------------- acting
    def _sample(self, prob):
        b = prob.shape[0]
        u = torch.from_numpy(np.random.rand(b)).to(self.device)       # special.py:24
        acts = torch.empty(b, dtype=torch.uint8, device=self.device)
        _lib.sample_categorical(prob.contiguous(), u, acts)
        return acts

    def get_actions(self, observations):
        prob, value = self.prob_value(observations)
        return self._sample(prob), dict(prob=prob, value=value)

    def get_action(self, observation, deterministic=False):
        prob, value = self.prob_value(observation[None])
        action = torch.argmax(prob[0]) if deterministic else self._sample(prob)[0]
        return action, dict(prob=prob[0], value=value[0])

    def reset(self, n_batch=None):
        pass

    def reset_one(self, idx):
        pass

    # ----------------------------------------------------------- parameters
    def get_params(self, trainable=True):
        return list(self.params)

    def _internal_arrays(self, flat=None):
        host = (self.flat_params if flat is None else flat).detach().cpu().numpy()
        return [host[o:o + int(np.prod(s))].reshape(s) for o, s in zip(self._offsets, self._shapes)]

    def get_param_values(self, trainable=True):
        """Flat fp32 vector in the reference's order and layout (host numpy)."""
        return self.bucket_to_reference(self.flat_params)

    def bucket_to_reference(self, flat):
        """Any tensor laid out like the internal flat bucket (parameters, gradients, optimiser
        slots) -> the reference's parameter-vector order and layout (host numpy)."""
        arr = self._internal_arrays(flat)
        out, k = [], 0
        for i in range(self._n_conv):
            if i == 0 and self._u8:                                         # already (K, C, kh, kw)
                out += [arr[k][:, :, ::-1, ::-1], arr[k + 1]]
                k += 2
                continue
            w = arr[k][..., :self._c_in] if i == 0 else arr[k]             # drop the zero padding channels
            out += [w.transpose(0, 3, 1, 2)[:, :, ::-1, ::-1], arr[k + 1]]
            k += 2
        tail = self._hidden_to_reference(arr[k:k + self._n_hidden_internal])
        k += self._n_hidden_internal
        out += self._tail_to_flat(tail + self._head_to_reference(arr[k], arr[k + 1]))
        return np.concatenate([np.ascontiguousarray(x).reshape(-1) for x in out]).astype(np.float32)

    def _set_from_reference_arrays(self, ref):
        host = np.zeros(self._bucket_len, np.float32)
        internal, k = [], 0
        for i in range(self._n_conv):
            w = ref[k][:, :, ::-1, ::-1]
            if not (i == 0 and self._u8):
                w = w.transpose(0, 2, 3, 1)
            if i == 0 and not self._u8 and self._c_pad != self._c_in:
                w = np.concatenate([w, np.zeros(w.shape[:3] + (self._c_pad - self._c_in,), np.float32)], axis=3)
            internal += [w, ref[k + 1]]
            k += 2
        tail = self._tail_from_flat(list(ref[k:]))
        internal += self._hidden_to_internal(tail[:self._n_hidden_tensors])
        internal += self._head_to_internal(tail[self._n_hidden_tensors:])
        for o, s, a in zip(self._offsets, self._shapes, internal):
            assert tuple(a.shape) == tuple(s), (a.shape, s)
            host[o:o + a.size] = np.ascontiguousarray(a).reshape(-1)
        self.flat_params.copy_(torch.from_numpy(host))

    def set_param_values(self, flat, trainable=True):
        flat = np.asarray(flat, np.float32)
        assert flat.size == self.n_params
        ref, pos = [], 0
        for s in self._ref_shapes:
            n = int(np.prod(s))
            ref.append(flat[pos:pos + n].reshape(s))
            pos += n
        self._set_from_reference_arrays(ref)
