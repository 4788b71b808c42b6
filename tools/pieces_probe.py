"""arl_conv_pieces: what reading / writing bf16 pieces costs or saves per launch at the spec-1 layer shapes
(event-timed, 20 launches between one pair of events).  usage: python tools/pieces_probe.py [batch]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from accel_rl_amd import _lib

DEV = "cuda:0"


def ev(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    ws = _lib.conv_workspace(DEV)
    _lib.load().arl_conv_tile_choice(int(os.environ.get("ARL_TILE_CHOICE", "0")))      # 3: gathered operand through LDS
    gen = torch.Generator(device=DEV).manual_seed(1)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=gen)                 # noqa: E731
    layers = [("conv2", _lib.conv_geom(b, 25, 19, 32, 64, 4, 4, 2, 1, 1), (b, 25, 19, 32), (64, 4, 4, 32), (b, 12, 9, 64)),
              ("conv3", _lib.conv_geom(b, 12, 9, 64, 64, 3, 3, 1, 1, 1), (b, 12, 9, 64), (64, 3, 3, 64), (b, 12, 9, 64)),
              ("dense", _lib.dense_geom(b, 6912, 512), (b, 6912), (512, 6912), (b, 512))]
    for name, g, xs, wsh, ys in layers:
        x, w, y, dy, dx = rnd(*xs).relu(), rnd(*wsh) * 0.05, torch.empty(*ys, device=DEV), rnd(*ys), torch.empty(*xs, device=DEV)
        px, py, pdy, pdx = (_lib.pieces_like(t) for t in (x, y, dy, dx))
        for op, caps_op in (("fwd", _lib.PIECES_FWD), ("dgrad", _lib.PIECES_DGRAD)):
            caps = _lib.conv_pieces_supported(g, caps_op)
            for pin in (False, True):
                for pout in (False, True):
                    if (pin and not caps & _lib.PIECES_IN) or (pout and not caps & _lib.PIECES_OUT):
                        continue

                    def fn():
                        if op == "fwd":
                            if pin or pout:
                                _lib.conv_pieces(px if pin else None, py if pout else None)
                            _lib.conv2d_fwd(x, w, None, y, g, True, ws)
                        else:
                            if pin or pout:
                                _lib.conv_pieces(pdy if pin else None, pdx if pout else None)
                            _lib.conv2d_bwd_data(dy, w, x, dx, g)
                    print("%-6s %-6s pieces in %d out %d   %7.1f us" % (name, op, pin, pout, ev(fn)), flush=True)
        _lib.conv_pieces(None, None)


if __name__ == "__main__":
    main()
