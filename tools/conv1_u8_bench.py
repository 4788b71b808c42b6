"""Convolution 1 from the u8 observations (arl_conv2d_u8_*) against the route it replaces
(arl_gather_scale_obs_nhwc + the NHWC kernels): microseconds per call, HIP events.
usage: python tools/conv1_u8_bench.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accel_rl_amd import _lib  # noqa: E402

DEV = "cuda:0"
SCALE = float(np.float32(1. / 255.))


def timeit(fn, reps=200):
    for _ in range(20):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    _lib.load()
    for rows, b, k in ((1280, 512, 32), (256, None, 32), (128, None, 32), (5120, None, 16), (1024, None, 16)):
        obs = torch.randint(0, 256, (rows, 4, 104, 80), device=DEV, dtype=torch.int32).to(torch.uint8)
        idx = None if b is None else torch.randperm(rows, device=DEV)[:b].to(torch.int32)
        n = rows if b is None else b
        geom = _lib.conv_geom(n, 104, 80, 4, k, 8, 8, 4, 0, 0)
        w = torch.randn(k, 4, 8, 8, device=DEV) / 16
        w_hwc = w.permute(0, 2, 3, 1).contiguous()
        bias = torch.zeros(k, device=DEV)
        y = torch.empty(n, 25, 19, k, device=DEV)
        x = torch.empty((n, 4, 104, 80), device=DEV, memory_format=torch.channels_last)
        dy = torch.randn(n, 25, 19, k, device=DEV)
        dw, db = torch.empty_like(w), torch.empty(k, device=DEV)
        ws, ws2 = _lib.conv_workspace(DEV), _lib.conv_workspace(DEV)
        folds = _lib.FoldList()
        t_g = timeit(lambda: _lib.gather_scale_obs_nhwc(obs, idx, x, SCALE))
        t_f = timeit(lambda: _lib.conv2d_fwd(x, w_hwc, bias, y, geom, True, ws))
        t_f8 = timeit(lambda: _lib.conv2d_u8_fwd(obs, idx, SCALE, w, bias, y, geom, True))

        def wg():
            folds.conv2d_bwd_weight(dy, x, dw, geom, ws2, dbias=db)
            folds.run()

        def wg8():
            folds.conv2d_u8_bwd_weight(dy, obs, idx, SCALE, dw, geom, ws2, dbias=db)
            folds.run()
        t_w, t_w8 = timeit(wg), timeit(wg8)
        print("rows %5d batch %5d filters %2d | gather %6.1f  fwd %6.1f  u8 fwd %6.1f | wgrad+fold %6.1f  u8 %6.1f  (us)"
              % (rows, n, k, t_g, t_f, t_f8, t_w, t_w8), flush=True)


if __name__ == "__main__":
    main()
