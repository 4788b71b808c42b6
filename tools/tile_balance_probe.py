"""How the 64-column forward kernels fill the chip: launch time of conv 2 / conv 3 forward at batches whose 128-row tile
counts are 256 (one workgroup per CU), 432 (the PPO minibatch: 1.69 per CU), 512 (two per CU), 768 (three) -- in-graph
timing, 20 launches per graph.  If T(512 tiles) ~ T(432 tiles), a balanced schedule of the 432 would take 432 / 512 of it.
usage: python tools/tile_balance_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from accel_rl_amd import _lib
from bench import graph_time_ms

DEV = "cuda:0"
LAYERS = dict(conv2=(25, 19, 32, 64, 4, 2, 1), conv3=(12, 9, 64, 64, 3, 1, 1))


def main():
    _lib.load()
    for name, (h, w, c, k, ks, st, p) in LAYERS.items():
        for b in (152, 228, 303, 304, 400, 512, 606, 607, 760, 910, 1024):
            geom = _lib.conv_geom(b, h, w, c, k, ks, ks, st, p, p)
            ho, wo = _lib.conv_out_hw(geom)
            ws = _lib.conv_workspace(DEV)
            x = torch.randn(b, h, w, c, device=DEV).relu()
            wt = torch.randn(k, ks, ks, c, device=DEV) / np.sqrt(ks * ks * c)
            bias = torch.randn(k, device=DEV)
            y = torch.empty(b, ho, wo, k, device=DEV)
            ms = graph_time_ms(lambda: _lib.conv2d_fwd(x, wt, bias, y, geom, True, ws))
            rows = b * ho * wo
            tiles = (rows + 127) // 128
            print("%s fwd B=%4d rows=%6d tiles(128)=%4d (%.2f per CU): %6.2f us  %.3f us per tile-round" %
                  (name, b, rows, tiles, tiles / 256.0, ms * 1e3, ms * 1e3 / np.ceil(tiles / 256.0)), flush=True)


main()
