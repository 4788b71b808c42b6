"""Feed-forward Atari CNN policy: conv stack -> dense -> (softmax pi, linear V).

Mirror of the reference's AtariCnnPolicy / PgCnn
(accel_rl/policies/pg/atari_cnn_policy.py:15-119, pg/networks/pg_cnn.py:11-119,
policies/layers.py:11-41) with PyTorch-ROCm as the dense-contraction engine (the
only MFMA-shaped stage of the path).  What is specific to this build:

* all trainable parameters live in ONE flat fp32 HBM bucket (`flat_params`) and
  all gradients in another (`flat_grads`); the nn.Parameters are views.  That is
  the vector the reference all-reduces (optimizers/util.py:35-39) and the HIP
  optimiser kernel updates in two launches (csrc/optim.hip).
* uint8 observations are converted by the gather/scale HIP kernel
  (x * 1/255, layers.py:22-41) instead of a host-side cast + H2D copy.
* action sampling runs on the device (csrc/batch_ops.hip) from numpy's uniforms.

Flat-vector convention of get/set_param_values = the reference's (Lasagne):
order conv_i.W, conv_i.b, hidden_i.W, hidden_i.b, output_pi.W, .b, output_v.W, .b
(pg_cnn.py:47-86,118-119); conv W (out,in,kh,kw) for a *flipped* (true)
convolution, dense W (in,out).  Internally torch uses correlation kernels and
(out,in) matrices; get/set convert.
"""
import numpy as np
import torch
import torch.nn.functional as F

from accel_rl_amd import _lib
from accel_rl_amd.distributions import Categorical
from accel_rl_amd.spaces import Discrete
from accel_rl_amd.util.seed import layer_rng


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


class AtariCnnPolicy(object):

    def __init__(self, conv_filters, conv_filter_sizes, conv_strides, conv_pads,
                 hidden_sizes=(), pixel_scale=255., initial_param_values=None):
        self.conv_filters, self.conv_filter_sizes = list(conv_filters), list(conv_filter_sizes)
        self.conv_strides, self.conv_pads = list(conv_strides), list(conv_pads)
        self.hidden_sizes = list(hidden_sizes)
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
        n_act = self.action_space.n
        self._obs_shape = (c, h, w)
        shapes, names, inits = [], [], []
        for i, (nf, sz, st, pad) in enumerate(zip(self.conv_filters, self.conv_filter_sizes,
                                                  self.conv_strides, self.conv_pads)):
            wshape = (nf, c, sz, sz)
            # stored as a correlation kernel: flip Lasagne's convolution filter spatially
            inits += [_glorot_uniform(wshape)[:, :, ::-1, ::-1].copy(), np.zeros(nf, np.float32)]
            shapes += [wshape, (nf,)]
            names += ["Conv%dW" % i, "Conv%db" % i]
            c = nf
            h = (h + 2 * pad[0] - sz) // st + 1
            w = (w + 2 * pad[1] - sz) // st + 1
        self._conv_out = (c, h, w)
        fan = c * h * w
        for i, hs in enumerate(self.hidden_sizes):
            inits += [_norm_c((fan, hs), 1.0).T.copy(), np.zeros(hs, np.float32)]
            shapes += [(hs, fan), (hs,)]
            names += ["FC%dW" % i, "FC%db" % i]
            fan = hs
        inits += [_norm_c((fan, n_act), 0.01).T.copy(), np.zeros(n_act, np.float32)]
        shapes += [(n_act, fan), (n_act,)]
        names += ["OutputW", "Outputb"]
        inits += [_norm_c((fan, 1), 1.0).T.copy(), np.zeros(1, np.float32)]
        shapes += [(1, fan), (1,)]
        names += ["OutputW", "Outputb"]
        self.param_short_names = names
        self._shapes = shapes
        sizes = [int(np.prod(s)) for s in shapes]
        # every tensor starts on a 16-byte boundary inside the bucket (float4 kernels)
        self._offsets, off = [], 0
        for n in sizes:
            self._offsets.append(off)
            off += (n + 3) // 4 * 4
        self.n_params = int(sum(sizes))
        self._bucket_len = off
        self.flat_params = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.flat_grads = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.params = []
        for o, n, s, init in zip(self._offsets, sizes, shapes, inits):
            p = torch.nn.Parameter(self.flat_params[o:o + n].view(s))
            p.data.copy_(torch.from_numpy(np.ascontiguousarray(init)))
            p.grad = self.flat_grads[o:o + n].view(s)
            self.params.append(p)
        self._n_conv = len(self.conv_filters)
        self._dist = Categorical(n_act)
        self._scale = float(np.float32(1. / self.pixel_scale))
        if self.initial_param_values is not None:
            self.set_param_values(self.initial_param_values)

    # -------------------------------------------------------------- forward
    def _scaled(self, obs_u8, idx=None):
        """u8 [B,C,H,W] (optionally gathered by idx) -> f32 * (1/pixel_scale)."""
        b = obs_u8.shape[0] if idx is None else idx.shape[0]
        key = (b, torch.is_grad_enabled())
        out = self._scratch.get(key)
        if out is None or torch.cuda.is_current_stream_capturing():
            out = torch.empty((b,) + self._obs_shape, dtype=torch.float32, device=self.device)
            if not torch.cuda.is_current_stream_capturing():
                self._scratch[key] = out
        _lib.gather_scale_obs(obs_u8, idx, out, self._scale)
        return out

    def forward(self, x):
        """f32 scaled pixels -> (prob [B,A], value [B])."""
        p = self.params
        for i in range(self._n_conv):
            x = F.relu(F.conv2d(x, p[2 * i], p[2 * i + 1], stride=self.conv_strides[i],
                                padding=tuple(self.conv_pads[i])))
        x = x.flatten(1)
        k = 2 * self._n_conv
        for _ in self.hidden_sizes:
            x = F.relu(F.linear(x, p[k], p[k + 1]))
            k += 2
        prob = torch.softmax(F.linear(x, p[k], p[k + 1]), dim=1)
        value = F.linear(x, p[k + 2], p[k + 3]).reshape(-1)
        return prob, value

    def prob_value(self, observations):
        """Batched inference on device uint8 observations (the sampler's hot call;
        reference: _f_prob_value, atari_cnn_policy.py:67,109)."""
        with torch.no_grad():
            return self.forward(self._scaled(observations))

    def dist_info_value_sym(self, obs_u8, idx=None):
        """Training-time forward (autograd) on a gathered minibatch."""
        prob, value = self.forward(self._scaled(obs_u8, idx))
        return dict(prob=prob), value

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
        if deterministic:
            action = torch.argmax(prob[0])
        else:
            action = self._sample(prob)[0]
        return action, dict(prob=prob[0], value=value[0])

    def reset(self, n_batch=None):
        pass

    def reset_one(self, idx):
        pass

    # ----------------------------------------------------------- parameters
    def get_params(self, trainable=True):
        return list(self.params)

    def _to_reference_layout(self, i, arr):
        if arr.ndim == 4:
            return arr[:, :, ::-1, ::-1]
        if arr.ndim == 2:
            return arr.T
        return arr

    def get_param_values(self, trainable=True):
        """Flat fp32 vector in the reference's order/layout (host numpy)."""
        host = self.flat_params.detach().cpu().numpy()
        parts = []
        for i, (o, s) in enumerate(zip(self._offsets, self._shapes)):
            n = int(np.prod(s))
            parts.append(np.ascontiguousarray(self._to_reference_layout(i, host[o:o + n].reshape(s))).reshape(-1))
        return np.concatenate(parts)

    def set_param_values(self, flat, trainable=True):
        flat = np.asarray(flat, np.float32)
        assert flat.size == self.n_params
        host = np.zeros(self._bucket_len, np.float32)
        pos = 0
        for i, (o, s) in enumerate(zip(self._offsets, self._shapes)):
            n = int(np.prod(s))
            ref_shape = s if len(s) != 2 else s[::-1]
            arr = self._to_reference_layout(i, flat[pos:pos + n].reshape(ref_shape))
            host[o:o + n] = np.ascontiguousarray(arr).reshape(-1)
            pos += n
        self.flat_params.copy_(torch.from_numpy(host))
