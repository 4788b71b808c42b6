"""Tile shape of the 64-column forward kernels (conv 2 / conv 3, split route) by batch: 128 x 64 (production), 128 x 32
(two column tiles per row tile) and 64 x 64 (both operands through LDS) through arl_dev_fwd_tile -- in-graph timing,
20 launches per graph, outputs compared bit for bit against the 128 x 64 tile.
usage: python tools/fwd_tile_probe.py [batch ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from accel_rl_amd import _lib
from bench import graph_time_ms

DEV = "cuda:0"
LAYERS = dict(conv2=(25, 19, 32, 64, 4, 2, 1), conv3=(12, 9, 64, 64, 3, 1, 1))
NAMES = {0: "128x64", 1: "128x32 x2", 2: "64x64"}


def main():
    lib = _lib.load()
    batches = [int(x) for x in sys.argv[1:]] or [32, 64, 128, 152, 256, 303, 400, 512, 606, 1024]
    for name, (h, w, c, k, ks, st, p) in LAYERS.items():
        for b in batches:
            geom = _lib.conv_geom(b, h, w, c, k, ks, ks, st, p, p)
            ho, wo = _lib.conv_out_hw(geom)
            ws = _lib.conv_workspace(DEV)
            x = torch.randn(b, h, w, c, device=DEV).relu()
            wt = torch.randn(k, ks, ks, c, device=DEV) / np.sqrt(ks * ks * c)
            bias = torch.randn(k, device=DEV)
            rows = b * ho * wo
            ref, line = None, []
            for v in (0, 1, 2):
                lib.arl_dev_fwd_tile(v)
                y = torch.empty(b, ho, wo, k, device=DEV)
                ms = graph_time_ms(lambda: _lib.conv2d_fwd(x, wt, bias, y, geom, True, ws))
                torch.cuda.synchronize()
                if ref is None:
                    ref = y.clone()
                line.append("%s %6.2f us%s" % (NAMES[v], ms * 1e3, "" if torch.equal(ref, y) else " (DIFFERS)"))
            lib.arl_dev_fwd_tile(-1)
            y = torch.empty(b, ho, wo, k, device=DEV)
            ms = graph_time_ms(lambda: _lib.conv2d_fwd(x, wt, bias, y, geom, True, ws))
            torch.cuda.synchronize()
            line.append("by size %6.2f us%s" % (ms * 1e3, "" if torch.equal(ref, y) else " (DIFFERS)"))
            print("%s fwd B=%4d rows=%6d tiles(128)=%4d: %s" % (name, b, rows, (rows + 127) // 128, "   ".join(line)), flush=True)


main()
