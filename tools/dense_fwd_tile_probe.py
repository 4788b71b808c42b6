"""Unsplit forward of the >= 128-column dense layers: 128 x 128 tiles (arl_dev_fwd_tile(7)) against 64 x 64 (6) and the
library's choice (-1: 64 x 64 when 0.85 x their CU fill beats the 128 x 128 tiles'), by shape -- alternating in-graph
timing (both orders), bit-for-bit comparison.   usage: python tools/dense_fwd_tile_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from accel_rl_amd import _lib
from bench import graph_time_ms

DEV = "cuda:0"
SHAPES = [(2560, 6912, 512), (4096, 6912, 512), (5120, 6912, 512), (8192, 6912, 512), (10240, 6912, 512), (8192, 3456, 256),
          (10240, 3456, 256), (5120, 512, 512), (5120, 2816, 256), (20000, 256, 128), (5120, 3456, 256)]


def main():
    lib = _lib.load()
    ws = _lib.conv_workspace(DEV)
    print("rows x fan_in -> units   tiles(128)   128 x 128    64 x 64    chosen     (us per launch, median of 8 alternations)")
    for b, c, k in SHAPES:
        geom = _lib.conv_geom(b, 1, 1, c, k, 1, 1, 1, 0, 0)
        x = torch.randn(b, 1, 1, c, device=DEV).relu()
        wt = torch.randn(k, 1, 1, c, device=DEV) / np.sqrt(c)
        bias = torch.randn(k, device=DEV)
        ts, outs = {7: [], 6: [], -1: []}, {}
        for rep in range(8):
            for v in ((7, 6, -1) if rep % 2 == 0 else (-1, 6, 7)):
                lib.arl_dev_fwd_tile(v)
                y = torch.full((b, 1, 1, k), float("nan"), device=DEV)
                ts[v].append(graph_time_ms(lambda: _lib.conv2d_fwd(x, wt, bias, y, geom, True, ws)) * 1e3)
                torch.cuda.synchronize()
                outs[v] = y
        lib.arl_dev_fwd_tile(-1)
        same = torch.equal(outs[7], outs[6]) and torch.equal(outs[7], outs[-1])
        t128 = ((b + 127) // 128) * ((k + 127) // 128)
        print("%6d x %5d -> %4d   %6d   %9.2f  %9.2f  %9.2f   %s" % (b, c, k, t128, np.median(ts[7]), np.median(ts[6]),
              np.median(ts[-1]), "bit-identical" if same else "(split shapes: forced values do not apply)"), flush=True)


main()
