"""Event-timed fp32 MFMA conv / dense kernels at the workload's shapes, next to
PyTorch-ROCm (MIOpen / hipBLASLt) on the same tensors.  usage: python tools/conv_bench.py [batch]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from accel_rl_amd import _lib

DEV = "cuda:0"
aten = torch.ops.aten


def ev(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    for i in range(reps):
        s[i].record(); fn(); e[i].record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in zip(s, e)])) * 1e3


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    ours_only = len(sys.argv) > 2
    layers = [("conv1", 104, 80, 4, 32, 8, 4, 0), ("conv2", 25, 19, 32, 64, 4, 2, 1),
              ("conv3", 12, 9, 64, 64, 3, 1, 1), ("dense", 1, 1, 6912, 512, 1, 1, 0)]
    ws = _lib.conv_workspace(DEV)
    print("batch %d; us per call; TF/s = 2*MACs/time" % b)
    for name, h, w, c, k, ks, st, p in layers:
        geom = _lib.conv_geom(b, h, w, c, k, ks, ks, st, p, p)
        ho, wo = _lib.conv_out_hw(geom)
        x = torch.randn(b, h, w, c, device=DEV)
        wt = torch.randn(k, ks, ks, c, device=DEV) * 0.05
        bias = torch.randn(k, device=DEV)
        y = torch.empty(b, ho, wo, k, device=DEV)
        dy = torch.randn(b, ho, wo, k, device=DEV)
        dx = torch.empty_like(x)
        dw = torch.empty_like(wt)
        flops = 2.0 * b * ho * wo * k * ks * ks * c
        xt = x.permute(0, 3, 1, 2)          # logical NCHW, channels-last memory
        wtt = wt.permute(0, 3, 1, 2)
        dyt = dy.permute(0, 3, 1, 2)
        t_f = ev(lambda: _lib.conv2d_fwd(x, wt, bias, y, geom, True, ws))
        t_d = ev(lambda: _lib.conv2d_bwd_data(dy, wt, None, dx, geom))
        t_w = ev(lambda: _lib.conv2d_bwd_weight(dy, x, dw, geom, ws))
        if ours_only:
            r_f = r_d = r_w = float("nan")
        elif name == "dense":
            x2, w2, dy2 = x.view(b, c), wt.view(k, c), dy.view(b, k)
            r_f = ev(lambda: torch.mm(x2, w2.t()))
            r_d = ev(lambda: torch.mm(dy2, w2))
            r_w = ev(lambda: torch.mm(dy2.t(), x2))
        else:
            r_f = ev(lambda: F.conv2d(xt, wtt, None, stride=st, padding=p))
            r_d = ev(lambda: aten.convolution_backward(dyt, xt, wtt, None, [st, st], [p, p], [1, 1], False, [0, 0], 1, [True, False, False]))
            r_w = ev(lambda: aten.convolution_backward(dyt, xt, wtt, None, [st, st], [p, p], [1, 1], False, [0, 0], 1, [False, True, False]))
        for tag, t, r in (("fwd", t_f, r_f), ("dgrad", t_d, r_d), ("wgrad", t_w, r_w)):
            print("%-6s %-6s ours %8.1f us (%6.1f TF/s)   torch %8.1f us (%6.1f TF/s)" %
                  (name, tag, t, flops / t / 1e6, r, flops / r / 1e6))


if __name__ == "__main__":
    main()
