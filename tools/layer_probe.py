"""GPU probe: per-layer conv fwd / bwd (dgrad+wgrad) and FC GEMM times, NCHW vs channels_last."""
import os, sys, time
import torch, torch.nn.functional as F
dev = "cuda:0"
aten = torch.ops.aten
def ev(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
print("PYTORCH_MIOPEN_SUGGEST_NHWC =", os.environ.get("PYTORCH_MIOPEN_SUGGEST_NHWC"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
layers = [("conv1", 4, 32, 8, 4, 0, (104, 80)), ("conv2", 32, 64, 4, 2, 1, (25, 19)), ("conv3", 64, 64, 3, 1, 1, (12, 9))]
for cl in (False, True):
    fmt = torch.channels_last if cl else torch.contiguous_format
    for name, ci, co, k, s, p, hw in layers:
        x = torch.randn(B, ci, *hw, device=dev).contiguous(memory_format=fmt)
        w = (torch.randn(co, ci, k, k, device=dev) * 0.05).contiguous(memory_format=fmt)
        y = F.conv2d(x, w, None, s, p)
        gy = torch.randn_like(y)
        t_f = ev(lambda: F.conv2d(x, w, None, s, p))
        need_dx = name != "conv1"
        t_b = ev(lambda: aten.convolution_backward(gy, x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [need_dx, True, False]))
        print("%s cl=%s out=%s strides=%s  fwd %.1f us  bwd(%s) %.1f us" % (name, cl, tuple(y.shape), y.stride(), t_f, "dx+dw" if need_dx else "dw", t_b), flush=True)
a = torch.randn(B, 6912, device=dev); w = torch.randn(512, 6912, device=dev) * 0.01; g = torch.randn(B, 512, device=dev)
print("fc fwd %.1f us, dW %.1f us, dx %.1f us" % (ev(lambda: a @ w.t()), ev(lambda: g.t() @ a), ev(lambda: g @ w)))
h = torch.randn(B, 512, device=dev); wh = torch.randn(5, 512, device=dev); go = torch.randn(B, 5, device=dev)
print("head fwd %.1f us, dW %.1f us, dh %.1f us" % (ev(lambda: h @ wh.t()), ev(lambda: go.t() @ h), ev(lambda: go @ wh)))
y = torch.randn(B, 32, 25, 19, device=dev); b = torch.randn(32, device=dev)
print("torch relu_(y+b) conv1-size: %.1f us; sum(0,2,3): %.1f us; threshold_bwd: %.1f us" % (
    ev(lambda: torch.relu_(y.add_(b.view(1, -1, 1, 1)))), ev(lambda: y.sum((0, 2, 3))), ev(lambda: aten.threshold_backward(y, y, 0))))
