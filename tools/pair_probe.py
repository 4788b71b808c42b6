"""The dense layer's paired data + weight gradient launch (arl_conv2d_bwd_pair) under the tile choices of the split
kernels: event-timed, and compared bit for bit with choice 0.  usage: python tools/pair_probe.py [batch] [choices...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from accel_rl_amd import _lib

DEV = "cuda:0"


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    choices = [int(c) for c in sys.argv[2:]] or [6, 0]
    lib = _lib.load()
    gen = torch.Generator(device=DEV).manual_seed(2)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=gen)                 # noqa: E731
    g = _lib.dense_geom(b, 6912, 512)
    dh, w, y3 = rnd(b, 512), rnd(512, 6912) * 0.02, rnd(b, 6912).relu()
    ws = _lib.conv_workspace(DEV)
    ref = None
    for c in choices:
        lib.arl_conv_tile_choice(c)
        d3, dw = torch.empty_like(y3), torch.empty_like(w)

        def fn():
            folds = _lib.FoldList()
            folds.conv2d_bwd_pair(dh, w, y3, d3, y3, dw, g, ws)
            folds.run()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        same = "" if ref is None else "  bit-identical to choice %d: %s" % (choices[0], torch.equal(d3, ref[0]) and torch.equal(dw, ref[1]))
        if ref is None:
            ref = (d3.clone(), dw.clone())
        print("dense pair, batch %d, tile choice %d: %.1f us%s" % (b, c, s.elapsed_time(e) * 1e3 / 20, same), flush=True)
    lib.arl_conv_tile_choice(0)


if __name__ == "__main__":
    main()
