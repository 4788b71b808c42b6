"""arl_conv_geom.route: the three routes of the fp32 contractions (0 = fp32 MFMA chain, 6 / 9 = bf16-split products)
at the spec-1 layer shapes -- error of every kernel against a float64 reference on the same inputs, and event-timed
launch durations at the PPO minibatch.  usage: python tools/split_check.py [timing batch] [accuracy batch]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from accel_rl_amd import _lib

DEV = "cuda:0"
MODES = tuple(int(m) for m in os.environ.get("ARL_MODES", "0,6,9").split(","))
LAYERS = [("conv1", 104, 80, 4, 32, 8, 4, 0), ("conv2", 25, 19, 32, 64, 4, 2, 1),
          ("conv3", 12, 9, 64, 64, 3, 1, 1), ("dense", 1, 1, 6912, 512, 1, 1, 0)]


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


def err(got, want):
    d = (got.double() - want).abs()
    return d.max().item() / max(want.abs().max().item(), 1e-30), (d.pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()


def run(b, timing):
    lib = _lib.load()
    ws = _lib.conv_workspace(DEV)
    gen = torch.Generator(device=DEV).manual_seed(3)
    for name, h, w, c, k, ks, st, p in LAYERS:
        geom = _lib.conv_geom(b, h, w, c, k, ks, ks, st, p, p)
        ho, wo = _lib.conv_out_hw(geom)
        x = torch.randn(b, h, w, c, device=DEV, generator=gen).relu()
        wt = torch.randn(k, ks, ks, c, device=DEV, generator=gen) / np.sqrt(ks * ks * c)
        bias = torch.randn(k, device=DEV, generator=gen)
        dy = torch.randn(b, ho, wo, k, device=DEV, generator=gen)
        y, dx, dw = torch.empty(b, ho, wo, k, device=DEV), torch.empty_like(x), torch.empty_like(wt)
        obs = torch.randint(0, 256, (b, c, h, w), device=DEV, dtype=torch.uint8, generator=gen) if name == "conv1" else None
        w8 = wt.permute(0, 3, 1, 2).contiguous() if obs is not None else None
        dw8 = torch.empty_like(w8) if obs is not None else None
        if not timing:
            xd, wd, dyd = x.double().permute(0, 3, 1, 2), wt.double().permute(0, 3, 1, 2), dy.double().permute(0, 3, 1, 2)
            xr, wr = xd.detach().requires_grad_(), wd.detach().requires_grad_()
            out = F.conv2d(xr, wr, None, stride=st, padding=p)
            gx, gw = torch.autograd.grad(out, (xr, wr), dyd)
            ref = dict(fwd=(out + bias.double().view(1, -1, 1, 1)).permute(0, 2, 3, 1).detach(), dgrad=gx.permute(0, 2, 3, 1), wgrad=gw.permute(0, 2, 3, 1))
            if obs is not None:
                o8 = obs.double() / 255.0
                o8r, w8r = o8.detach().requires_grad_(), w8.double().detach().requires_grad_()
                out8 = F.conv2d(o8r, w8r, None, stride=st)
                ref["u8fwd"] = (out8 + bias.double().view(1, -1, 1, 1)).permute(0, 2, 3, 1).detach()
                ref["u8wgrad"] = torch.autograd.grad(out8, w8r, dyd)[0]
        for mode in MODES:
            _lib.set_conv_precision(mode)
            geom = _lib.with_route(geom)
            ops = dict(fwd=lambda: _lib.conv2d_fwd(x, wt, bias, y, geom, False, ws),
                       dgrad=lambda: _lib.conv2d_bwd_data(dy, wt, None, dx, geom),
                       wgrad=lambda: _lib.conv2d_bwd_weight(dy, x, dw, geom, ws))
            outs = dict(fwd=y, dgrad=dx, wgrad=dw)
            if obs is not None:
                db = torch.empty(k, device=DEV)
                folds = _lib.FoldList()

                def u8w():
                    folds.conv2d_u8_bwd_weight(dy, obs, None, 1.0 / 255.0, dw8, geom, ws, dbias=db)
                    folds.run()
                ops["u8fwd"] = lambda: _lib.conv2d_u8_fwd(obs, None, 1.0 / 255.0, w8, bias, y, geom, False)
                ops["u8wgrad"] = u8w
                outs["u8fwd"], outs["u8wgrad"] = y, dw8
            for op, fn in ops.items():
                if timing:
                    print("%-6s %-8s mode %d  %7.1f us" % (name, op, mode, ev(fn)), flush=True)
                else:
                    outs[op].fill_(float("nan"))
                    fn()
                    torch.cuda.synchronize()
                    mx, rms = err(outs[op], ref[op])
                    print("%-6s %-8s mode %d  max err / max|ref| %.3e   rms err / rms ref %.3e" % (name, op, mode, mx, rms), flush=True)
    _lib.set_conv_precision(9)


if __name__ == "__main__":
    tb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    ab = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    print("== accuracy against float64, batch %d" % ab)
    run(ab, False)
    print("== timing, batch %d (20 launches between one pair of events)" % tb)
    run(tb, True)
