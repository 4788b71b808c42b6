"""GPU probe: spec-1 CNN forward / forward+backward time under MIOpen settings."""
import sys, time, os
import torch, torch.nn.functional as F
dev = "cuda:0"
def net_params(cl):
    torch.manual_seed(0)
    ws = [torch.randn(32, 4, 8, 8), torch.randn(64, 32, 4, 4), torch.randn(64, 64, 3, 3)]
    ws = [(w * 0.05).to(dev).requires_grad_() for w in ws]
    if cl:
        ws = [w.detach().contiguous(memory_format=torch.channels_last).requires_grad_() for w in ws]
    bs = [torch.zeros(n, device=dev, requires_grad=True) for n in (32, 64, 64)]
    fc = (torch.randn(512, 6912, device=dev) * 0.01).requires_grad_()
    fb = torch.zeros(512, device=dev, requires_grad=True)
    pi = (torch.randn(4, 512, device=dev) * 0.01).requires_grad_()
    vv = (torch.randn(1, 512, device=dev) * 0.01).requires_grad_()
    return ws, bs, fc, fb, pi, vv
def fwd(x, P):
    ws, bs, fc, fb, pi, vv = P
    x = F.relu(F.conv2d(x, ws[0], bs[0], stride=4))
    x = F.relu(F.conv2d(x, ws[1], bs[1], stride=2, padding=1))
    x = F.relu(F.conv2d(x, ws[2], bs[2], stride=1, padding=1))
    x = F.relu(F.linear(x.flatten(1), fc, fb))
    return torch.softmax(F.linear(x, pi), 1), F.linear(x, vv)
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    for cl in (False, True):
        P = net_params(cl)
        for B, train in ((256, False), (512, True)):
            x = torch.rand(B, 4, 104, 80, device=dev)
            if cl: x = x.contiguous(memory_format=torch.channels_last)
            if train:
                def f():
                    p, v = fwd(x, P); (p.sum() + v.sum()).backward()
            else:
                def f():
                    with torch.no_grad(): fwd(x, P)
            print("benchmark=%s channels_last=%s B=%d %s: %.3f ms" % (bench, cl, B, "fwd+bwd" if train else "fwd", timeit(f)), flush=True)
