"""MFMA implicit-GEMM convolution / dense kernels (csrc/mfma_conv*.hip; every route of arl_conv_geom.route) against
plain PyTorch fp32 on the same inputs.  Floating point, so a tolerance: the kernels
accumulate in fp32 in k order (bitwise an fmaf chain); torch's reference reduces in a
different order, so results agree to fp32 round-off of the reduction:
|got - want| <= 2e-5 * sqrt(K_red) * max|want|  (observed ~1e-6 * max|want|).
Run-to-run the kernels must be BIT-identical (no atomics, fixed-order folds)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

#        batch  H    W   C   K  k  s  p
CASES = [(37, 104, 80, 4, 32, 8, 4, 0),      # spec-1 conv 1 (ragged batch)
         (64, 104, 80, 4, 16, 8, 4, 0),      # spec-0 conv 1
         (33, 25, 19, 32, 64, 4, 2, 1),      # spec-1 conv 2 (stride-2 parity classes, odd image)
         (16, 25, 19, 16, 32, 4, 2, 1),      # spec-0 conv 2
         (50, 12, 9, 64, 64, 3, 1, 1),       # spec-1 conv 3
         (1, 12, 9, 64, 64, 3, 1, 1),        # single image
         (512, 1, 1, 6912, 512, 1, 1, 0),    # spec-1 dense at the PPO minibatch (split-K forward)
         (256, 1, 1, 6912, 512, 1, 1, 0),    # ... at the rollout batch (9 k-tiles per split: odd)
         (256, 1, 1, 3840, 512, 1, 1, 0),    # 5 k-tiles per split (the two-tile unrolled loop's odd tail)
         (80, 1, 1, 3456, 256, 1, 1, 0),     # spec-0 dense, ragged rows
         (5120, 1, 1, 512, 128, 1, 1, 0),    # wide batch: 40 tiles of 128x128, reduction split 4 ways
         (1024, 1, 1, 2816, 256, 1, 1, 0),   # spec-0 dense at a mid-size batch: 128x128 tiles + split-K
         (20000, 1, 1, 256, 128, 1, 1, 0),   # enough tiles: no split
         (32, 1, 1, 256, 1152, 1, 1, 0)]     # C51 head at the DQN batch: 4 output tiles, data gradient splits its reduction


def _mk(case, seed=0):
    from accel_rl_amd import _lib
    b, h, w, c, k, ks, st, p = case
    gen = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn(b, h, w, c, device=DEV, generator=gen)
    x = torch.where(torch.rand(x.shape, device=DEV, generator=gen) < 0.3, torch.zeros_like(x), x)
    wt = torch.randn(k, ks, ks, c, device=DEV, generator=gen) / np.sqrt(ks * ks * c)
    bias = torch.randn(k, device=DEV, generator=gen)
    geom = _lib.conv_geom(b, h, w, c, k, ks, ks, st, p, p)
    ws = _lib.conv_workspace(DEV)
    return x, wt, bias, geom, ws


def _tol(want, k_red):
    return 2e-5 * np.sqrt(k_red) * max(want.abs().max().item(), 1e-6)


@pytest.fixture(params=["fast", "fast_split6", "fast_fp32", "generic"])
def path(request):
    """Both kernel families: the scalar-addressed fast path (taken whenever a k-tile of 32 stays inside
    one filter row) and the generic fallback (any multiple-of-4 channel count, any K); on the fast path the three
    routes of arl_conv_geom.route (default: nine bf16-split products; six; the fp32 MFMA chain)."""
    from accel_rl_amd import _lib
    lib = _lib.load()
    lib.arl_dev_conv_force_generic(1 if request.param == "generic" else 0)
    _lib.set_conv_precision({"fast": 9, "fast_split6": 6}.get(request.param, 0))
    yield request.param
    lib.arl_dev_conv_force_generic(0)
    _lib.set_conv_precision(9)


ODD_CASES = [(9, 20, 14, 12, 20, 3, 1, 1),      # channels 12 / 20: no power-of-two anywhere -> generic kernels
             (7, 17, 13, 8, 24, 5, 1, 2),       # 5x5, pad 2
             (11, 1, 1, 100, 36, 1, 1, 0)]      # dense with K = 100 (not a multiple of the k-tile)


@pytest.mark.parametrize("case", ODD_CASES)
def test_odd_shapes_take_the_generic_kernels(case):
    from accel_rl_amd import _lib
    b, h, w, c, k, ks, st, p = case
    x, wt, bias, geom, ws = _mk(case, seed=2)
    ho, wo = _lib.conv_out_hw(geom)
    y = torch.full((b, ho, wo, k), float("nan"), device=DEV)
    _lib.conv2d_fwd(x, wt, bias, y, geom, True, ws)
    want = F.relu(F.conv2d(x.permute(0, 3, 1, 2), wt.permute(0, 3, 1, 2), bias, stride=st, padding=p)).permute(0, 2, 3, 1)
    assert (y - want).abs().max().item() <= _tol(want, ks * ks * c)
    dy = torch.randn(b, ho, wo, k, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    xr = x.permute(0, 3, 1, 2).detach().requires_grad_()
    wr = wt.permute(0, 3, 1, 2).detach().requires_grad_()
    gx, gw = torch.autograd.grad(F.conv2d(xr, wr, None, stride=st, padding=p), (xr, wr), dy.permute(0, 3, 1, 2))
    dx, dw = torch.full_like(x, float("nan")), torch.full_like(wt, float("nan"))
    _lib.conv2d_bwd_data(dy, wt, None, dx, geom)
    _lib.conv2d_bwd_weight(dy, x, dw, geom, ws)
    assert (dx - gx.permute(0, 2, 3, 1)).abs().max().item() <= _tol(gx, ks * ks * k)
    assert (dw - gw.permute(0, 2, 3, 1)).abs().max().item() <= _tol(gw, b * ho * wo)


@pytest.mark.parametrize("case", CASES)
def test_forward_matches_torch(case, path):
    from accel_rl_amd import _lib
    b, h, w, c, k, ks, st, p = case
    x, wt, bias, geom, ws = _mk(case)
    ho, wo = _lib.conv_out_hw(geom)
    for relu in (True, False):
        y = torch.full((b, ho, wo, k), float("nan"), device=DEV)
        _lib.conv2d_fwd(x, wt, bias, y, geom, relu, ws)
        want = F.conv2d(x.permute(0, 3, 1, 2), wt.permute(0, 3, 1, 2), bias, stride=st, padding=p)
        want = (F.relu(want) if relu else want).permute(0, 2, 3, 1)
        assert torch.isfinite(y).all()
        assert (y - want).abs().max().item() <= _tol(want, ks * ks * c), (case, relu)
        y2 = torch.empty_like(y)
        _lib.conv2d_fwd(x, wt, bias, y2, geom, relu, ws)
        assert torch.equal(y, y2)
    y = torch.empty((b, ho, wo, k), device=DEV)
    _lib.conv2d_fwd(x, wt, None, y, geom, False, ws)
    want = F.conv2d(x.permute(0, 3, 1, 2), wt.permute(0, 3, 1, 2), None, stride=st, padding=p).permute(0, 2, 3, 1)
    assert (y - want).abs().max().item() <= _tol(want, ks * ks * c)


@pytest.mark.parametrize("case", CASES)
def test_backward_matches_autograd(case, path):
    from accel_rl_amd import _lib
    b, h, w, c, k, ks, st, p = case
    x, wt, bias, geom, ws = _mk(case, seed=1)
    ho, wo = _lib.conv_out_hw(geom)
    gen = torch.Generator(device=DEV).manual_seed(7)
    dy = torch.randn(b, ho, wo, k, device=DEV, generator=gen)
    xr = x.permute(0, 3, 1, 2).detach().requires_grad_()
    wr = wt.permute(0, 3, 1, 2).detach().requires_grad_()
    out = F.conv2d(xr, wr, None, stride=st, padding=p)
    gx, gw = torch.autograd.grad(out, (xr, wr), dy.permute(0, 3, 1, 2))
    gx, gw = gx.permute(0, 2, 3, 1), gw.permute(0, 2, 3, 1)
    # ---- data gradient (with and without the fused rectifier mask)
    dx = torch.full((b, h, w, c), float("nan"), device=DEV)
    _lib.conv2d_bwd_data(dy, wt, None, dx, geom)
    assert torch.isfinite(dx).all()
    assert (dx - gx).abs().max().item() <= _tol(gx, (ks // st) ** 2 * k), case
    dxm = torch.full((b, h, w, c), float("nan"), device=DEV)
    _lib.conv2d_bwd_data(dy, wt, x, dxm, geom)
    assert torch.equal(dxm, torch.where(x > 0, dx, torch.zeros_like(dx)))
    # ---- weight gradient
    dw = torch.full((k, ks, ks, c), float("nan"), device=DEV)
    _lib.conv2d_bwd_weight(dy, x, dw, geom, ws)
    assert torch.isfinite(dw).all()
    assert (dw - gw).abs().max().item() <= _tol(gw, b * ho * wo), case
    dw2 = torch.empty_like(dw)
    _lib.conv2d_bwd_weight(dy, x, dw2, geom, ws)
    assert torch.equal(dw, dw2)                                 # split reduction is deterministic


@pytest.mark.parametrize("case", [(512, 104, 80, 4, 32, 8, 4, 0), (512, 25, 19, 32, 64, 4, 2, 1),
                                  (512, 12, 9, 64, 64, 3, 1, 1), (512, 1, 1, 6912, 512, 1, 1, 0),
                                  (5120, 12, 9, 64, 64, 3, 1, 1)])
def test_adjoint_identities_at_full_size(case):
    """Size-independent property at the PPO minibatch (B = 512) and the A2C batch (5120) of BASELINE
    configs 2-3: the three kernels are one bilinear form,
        <conv(x; w), dy> = <x, bwd_data(dy; w)> = <w, bwd_weight(dy; x)>,
    evaluated in fp64 from the fp32 outputs (no reference convolution needed at this size)."""
    from accel_rl_amd import _lib
    b, h, w, c, k, ks, st, p = case
    x, wt, bias, geom, ws = _mk(case, seed=3)
    ho, wo = _lib.conv_out_hw(geom)
    dy = torch.randn(b, ho, wo, k, device=DEV, generator=torch.Generator(device=DEV).manual_seed(9))
    y, dw = torch.empty_like(dy), torch.empty_like(wt)
    _lib.conv2d_fwd(x, wt, None, y, geom, False, ws)
    _lib.conv2d_bwd_weight(dy, x, dw, geom, ws)
    dot = lambda a, b_: torch.dot(a.double().reshape(-1), b_.double().reshape(-1)).item()     # noqa: E731
    ref = dot(y, dy)
    scale = np.sqrt(dot(y, y) * dot(dy, dy))
    assert abs(dot(wt, dw) - ref) <= 1e-5 * scale, (case, ref, dot(wt, dw))
    if ks % st == 0:
        dx = torch.empty_like(x)
        _lib.conv2d_bwd_data(dy, wt, None, dx, geom)
        assert abs(dot(x, dx) - ref) <= 1e-5 * scale, (case, ref, dot(x, dx))


@pytest.mark.parametrize("case", [c for c in CASES if c[5] % c[6] == 0] + ODD_CASES +
                         [(20, 12, 9, 128, 96, 3, 1, 1)])      # wide conv: shares a launch like the dense layers
@pytest.mark.parametrize("masked", [False, True])
def test_paired_backward_is_the_two_separate_calls(case, masked):
    """arl_conv2d_bwd_pair (one launch, deferred fold) == arl_conv2d_bwd_data + arl_conv2d_bwd_weight, bit for
    bit: the same tiles in the same order, only co-scheduled (or the documented fallbacks: separate launches for
    odd shapes; a split reduction for thin dense data gradients)."""
    from accel_rl_amd import _lib
    x, wt, bias, geom, ws = _mk(case, seed=3)
    ho, wo = _lib.conv_out_hw(geom)
    gen = torch.Generator(device=DEV).manual_seed(5)
    dy = torch.randn(case[0], ho, wo, case[4], device=DEV, generator=gen)
    mask = torch.randn(x.shape, device=DEV, generator=gen) if masked else None
    dx0, dw0 = torch.full_like(x, float("nan")), torch.full_like(wt, float("nan"))
    _lib.conv2d_bwd_data(dy, wt, mask, dx0, geom)
    _lib.conv2d_bwd_weight(dy, x, dw0, geom, ws)
    dx1, dw1 = torch.full_like(x, float("nan")), torch.full_like(wt, float("nan"))
    folds, ws2 = _lib.FoldList(), _lib.conv_workspace(DEV)
    folds.conv2d_bwd_pair(dy, wt, mask, dx1, x, dw1, geom, ws2)
    folds.run()
    assert torch.equal(dw0, dw1)
    # the data gradient too -- except for a stride-1 layer with few output tiles (dense layers at small batch),
    # whose reduction the paired call splits across workgroups (it has a workspace, the bare call has not)
    split_ok = case[6] == 1 and case[3] > 64 and x.numel() <= 1 << 20

    def same_dx(a, b):
        if torch.equal(a, b):
            return True
        return split_ok and (a - b).abs().max().item() <= 2e-6 * np.sqrt(dy.numel() / case[0]) * a.abs().max().item()
    assert same_dx(dx0, dx1)
    # ... and with the bias gradient's column sums riding along (fast kernels only; else "not produced")
    want_db = dy.reshape(-1, case[4]).double().sum(0)
    for paired in (True, False):
        dx2, dw2 = torch.full_like(x, float("nan")), torch.full_like(wt, float("nan"))
        db = torch.full((case[4],), float("nan"), device=DEV)
        if paired:
            done = folds.conv2d_bwd_pair(dy, wt, mask, dx2, x, dw2, geom, ws2, dbias=db)
        else:
            done = folds.conv2d_bwd_weight(dy, x, dw2, geom, ws2, dbias=db)
        folds.run()
        assert done == ((case[0] * ho * wo) % 32 == 0)       # the scalar-addressed kernels need whole 32-row k-tiles
        assert torch.equal(dw0, dw2) and (not paired or same_dx(dx0, dx2))
        if done:
            tol = 2e-5 * np.sqrt(dy.numel() / case[4]) * dy.abs().max().item()
            assert (db.double() - want_db).abs().max().item() <= tol
        else:
            assert torch.isnan(db).all()


def test_argument_errors():
    from accel_rl_amd import _lib
    x, wt, bias, geom, ws = _mk((2, 12, 9, 64, 64, 3, 1, 1))
    y = torch.empty(2, 12, 9, 64, device=DEV)
    bad = _lib.conv_geom(2, 12, 9, 6, 64, 3, 3, 1, 1, 1)        # channels not a multiple of 4
    with pytest.raises(RuntimeError):
        _lib.load().arl_conv2d_fwd(x.data_ptr(), wt.data_ptr(), None, y.data_ptr(), _lib.C.byref(bad), 0,
                                   ws.data_ptr(), None) and (_ for _ in ()).throw(RuntimeError("rc"))
    odd = _lib.conv_geom(2, 13, 9, 64, 64, 3, 3, 2, 1, 1)       # kernel 3, stride 2: unsupported data gradient
    rc = _lib.load().arl_conv2d_bwd_data(y.data_ptr(), wt.data_ptr(), None, None, x.data_ptr(), _lib.C.byref(odd), None, None, None)
    assert rc == -2 and b"stride" in _lib.load().arl_last_error()
    with pytest.raises(RuntimeError):
        _lib.conv2d_fwd(x.cpu(), wt, None, y, geom, False, ws) if False else _lib.ptr(x.cpu())


def test_fold_many_matches_separate_folds_and_checks_arguments():
    """arl_fold_many: several independent split folds in one launch == the per-tensor fold (same fixed order);
    items with splits <= 0 are skipped; bad items are refused."""
    from accel_rl_amd import _lib
    lib = _lib.load()
    gen = torch.Generator(device=DEV).manual_seed(2)
    items = (_lib.ArlFoldItem * 4)()
    parts, outs, wants = [], [], []
    for i, (splits, total) in enumerate([(7, 64), (33, 4), (1, 1028), (16, 20)]):
        p = torch.randn(splits, total, device=DEV, generator=gen)
        o = torch.full((total,), float("nan"), device=DEV)
        # reference order of fold_splits_kernel: lane zg sums splits zg, zg + 16, ...; then the 16 lanes in order
        lanes = [p[zg::16].double().float() for zg in range(16)]
        acc = []
        for l in lanes:
            t = torch.zeros(total, device=DEV)
            for row in l:
                t = t + row
            acc.append(t)
        w = acc[0]
        for t in acc[1:]:
            w = w + t
        parts.append(p); outs.append(o); wants.append(w)
        items[i].part, items[i].out, items[i].total, items[i].splits = p.data_ptr(), o.data_ptr(), total, splits
    items[3].splits = 0                                             # "already final": left alone
    assert lib.arl_fold_many(items, 4, None) == 0
    torch.cuda.synchronize()
    for i in range(3):
        assert torch.equal(outs[i], wants[i]), i
    assert torch.isnan(outs[3]).all()
    items[3].splits, items[3].total = 16, 18                        # not a multiple of 4
    assert lib.arl_fold_many(items, 4, None) < 0
    assert lib.arl_fold_many(items, _lib.FOLD_MAX_ITEMS + 1, None) < 0
    assert lib.arl_fold_many(None, 0, None) == 0


def _random_cases(n, seed):
    """Random geometries inside the kernels' contract (channels / filters multiples of 4, kernel size a
    multiple of the stride for the data gradient), biased towards the fast-path conditions (multiples of
    32) but including ragged batches, odd images, odd k-tile counts and split reductions."""
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        st = int(rs.choice([1, 1, 2, 4]))
        ks = int(st * rs.choice([1, 2, 3]) if st > 1 else rs.choice([1, 3, 5]))
        c = int(rs.choice([4, 8, 16, 32, 64, 96]))
        k = int(rs.choice([4, 16, 32, 64, 128]))
        h, w = int(rs.randint(ks, 30)), int(rs.randint(ks, 30))
        p = int(rs.randint(0, 2)) if ks > 1 else 0
        b = int(rs.choice([1, 7, 32, 64, 100]))
        if ks == 1 and rs.rand() < 0.5:                      # dense layers: long reductions, split K
            h = w = 1
            c = int(rs.choice([256, 1152, 3840, 4000]))
        out.append((b, h, w, c, k, ks, st, p))
    return out


@pytest.mark.parametrize("case", _random_cases(36, seed=20240930))
def test_random_geometries_against_torch_and_the_generic_kernels(case):
    """Forward, data gradient (masked) and weight gradient on random shapes: fp32 tolerance against
    PyTorch (same bound as above) for whichever kernel family the dispatcher picks, and against the
    generic kernels forced on the same inputs."""
    from accel_rl_amd import _lib
    b, h, w, c, k, ks, st, p = case
    x, wt, bias, geom, ws = _mk(case, seed=11)
    ho, wo = _lib.conv_out_hw(geom)
    gen = torch.Generator(device=DEV).manual_seed(13)
    dy = torch.randn(b, ho, wo, k, device=DEV, generator=gen)
    xr = x.permute(0, 3, 1, 2).detach().requires_grad_()
    wr = wt.permute(0, 3, 1, 2).detach().requires_grad_()
    out = F.conv2d(xr, wr, bias, stride=st, padding=p)
    gx, gw = torch.autograd.grad(out, (xr, wr), dy.permute(0, 3, 1, 2))
    want_y = F.relu(out.detach()).permute(0, 2, 3, 1)
    gx, gw = gx.permute(0, 2, 3, 1), gw.permute(0, 2, 3, 1)
    res = {}
    for generic in (0, 1):
        _lib.load().arl_dev_conv_force_generic(generic)
        try:
            y = torch.full((b, ho, wo, k), float("nan"), device=DEV)
            _lib.conv2d_fwd(x, wt, bias, y, geom, True, ws)
            dx = torch.full((b, h, w, c), float("nan"), device=DEV)
            _lib.conv2d_bwd_data(dy, wt, x, dx, geom)
            dw = torch.full((k, ks, ks, c), float("nan"), device=DEV)
            _lib.conv2d_bwd_weight(dy, x, dw, geom, ws)
        finally:
            _lib.load().arl_dev_conv_force_generic(0)
        assert (y - want_y).abs().max().item() <= _tol(want_y, ks * ks * c), (case, generic)
        want_dx = torch.where(x > 0, gx, torch.zeros_like(gx))
        assert (dx - want_dx).abs().max().item() <= _tol(gx, (ks // st) ** 2 * k), (case, generic)
        assert (dw - gw).abs().max().item() <= _tol(gw, b * ho * wo), (case, generic)
        res[generic] = (y, dx, dw)
    for a, g_ in zip(res[0], res[1]):                      # the two families agree to round-off too
        assert torch.allclose(a, g_, rtol=1e-4, atol=1e-4 * max(g_.abs().max().item(), 1e-6))


SPEC1 = [("conv1", 104, 80, 4, 32, 8, 4, 0), ("conv2", 25, 19, 32, 64, 4, 2, 1), ("conv3", 12, 9, 64, 64, 3, 1, 1),
         ("dense", 1, 1, 3456, 512, 1, 1, 0)]


@pytest.mark.parametrize("layer", SPEC1, ids=[c[0] for c in SPEC1])
def test_every_precision_route_is_fp32_accurate_against_float64(layer):
    """arl_conv_geom.route: the bf16-split routes (nine / six piece products accumulated in fp32 by the bf16 MFMAs) must be
    as close to the EXACT (float64) contraction as the fp32 MFMA chain is -- at the spec-1 layer shapes, forward, data and
    weight gradient, and conv 1 from u8 rows.  Bar: rms error <= 6e-7 of the rms of the exact result for every route
    (observed 0.7e-7 .. 4.1e-7, the fp32 chain the largest), and a split route at most 1.5x the fp32 chain's error + 3e-8.
    Also: every route is run-to-run bit-identical, and an unknown route is refused."""
    from accel_rl_amd import _lib
    name, h, w, c, k, ks, st, p = layer
    b = 48
    gen = torch.Generator(device=DEV).manual_seed(3)
    geom = _lib.conv_geom(b, h, w, c, k, ks, ks, st, p, p)
    ho, wo = _lib.conv_out_hw(geom)
    ws = _lib.conv_workspace(DEV)
    x = torch.randn(b, h, w, c, device=DEV, generator=gen).relu()
    wt = torch.randn(k, ks, ks, c, device=DEV, generator=gen) / np.sqrt(ks * ks * c)
    bias = torch.randn(k, device=DEV, generator=gen)
    dy = torch.randn(b, ho, wo, k, device=DEV, generator=gen)
    nchw = lambda t: t.double().permute(0, 3, 1, 2)                              # noqa: E731
    xr, wr = nchw(x).detach().requires_grad_(), nchw(wt).detach().requires_grad_()
    out = F.conv2d(xr, wr, None, stride=st, padding=p)
    gx, gw = torch.autograd.grad(out, (xr, wr), nchw(dy))
    ref = dict(fwd=(out + bias.double().view(1, -1, 1, 1)).permute(0, 2, 3, 1).detach(), dgrad=gx.permute(0, 2, 3, 1),
               wgrad=gw.permute(0, 2, 3, 1))
    obs = w8 = None
    if name == "conv1":
        obs = torch.randint(0, 256, (b, c, h, w), device=DEV, dtype=torch.int32, generator=gen).to(torch.uint8)
        w8 = wt.permute(0, 3, 1, 2).contiguous()
        o8r, w8r = (obs.double() / 255.0).requires_grad_(), w8.double().detach().requires_grad_()
        out8 = F.conv2d(o8r, w8r, None, stride=st)
        ref["u8fwd"] = (out8 + bias.double().view(1, -1, 1, 1)).permute(0, 2, 3, 1).detach()
        ref["u8wgrad"] = torch.autograd.grad(out8, w8r, nchw(dy))[0]

    def run_all(g):
        y, dx, dw = torch.empty(b, ho, wo, k, device=DEV), torch.empty_like(x), torch.empty_like(wt)
        _lib.conv2d_fwd(x, wt, bias, y, g, False, ws)
        _lib.conv2d_bwd_data(dy, wt, None, dx, g)
        _lib.conv2d_bwd_weight(dy, x, dw, g, ws)
        got = dict(fwd=y, dgrad=dx, wgrad=dw)
        if obs is not None:
            y8, dw8, db = torch.empty_like(y), torch.empty_like(w8), torch.empty(k, device=DEV)
            _lib.conv2d_u8_fwd(obs, None, 1.0 / 255.0, w8, bias, y8, g, False)
            folds = _lib.FoldList()
            folds.conv2d_u8_bwd_weight(dy, obs, None, 1.0 / 255.0, dw8, g, ws, dbias=db)
            folds.run()
            got.update(u8fwd=y8, u8wgrad=dw8)
        torch.cuda.synchronize()
        return got
    errs = {}
    for mode, route in ((0, _lib.ROUTE_FP32), (6, _lib.ROUTE_SPLIT6), (9, _lib.ROUTE_SPLIT9)):
        g = _lib.with_route(geom, route)
        got, again = run_all(g), run_all(g)
        for op in got:
            assert torch.equal(got[op], again[op]), (mode, op)
            errs[mode, op] = _rel_rms(got[op], ref[op])
            assert errs[mode, op] <= 6e-7, (mode, op, errs[mode, op])
    for mode in (6, 9):
        for op in ref:
            assert errs[mode, op] <= 1.5 * errs[0, op] + 3e-8, (mode, op, errs[mode, op], errs[0, op])
    y = torch.empty(b, ho, wo, k, device=DEV)
    with pytest.raises(RuntimeError, match="conv route"):
        _lib.conv2d_fwd(x, wt, bias, y, _lib.with_route(geom, 7), False, ws)
    with pytest.raises(ValueError):
        _lib.set_conv_precision(7)


def _rel_rms(got, want):
    return ((got.double() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()


PAIRED = [("conv2", 25, 19, 32, 64, 4, 2, 1), ("dense", 1, 1, 6912, 512, 1, 1, 0)]


@pytest.mark.parametrize("layer", PAIRED, ids=[c[0] for c in PAIRED])
def test_learner_batch_backward_is_fp32_accurate_against_float64(layer):
    """The float64 comparison at the LEARNER's batch (512 rows), where the routes take their own kernels (at 48 rows the
    small launches of conv 1 / the dense layers fall to one kernel whatever the route), through the call the learner makes:
    arl_conv2d_bwd_pair -- for the dense layer the one launch that holds data and weight gradient (bwd_pair_kernel, 12 % of
    a PPO step), for conv 2 its split kernels launched apart.  Same bars as above ('weak': float64 test does
    not reach every B = 512 kernel)."""
    from accel_rl_amd import _lib
    name, h, w, c, k, ks, st, p = layer
    b = 512
    gen = torch.Generator(device=DEV).manual_seed(17)
    geom = _lib.conv_geom(b, h, w, c, k, ks, ks, st, p, p)
    ho, wo = _lib.conv_out_hw(geom)
    x = torch.randn(b, h, w, c, device=DEV, generator=gen).relu()
    wt = torch.randn(k, ks, ks, c, device=DEV, generator=gen) / np.sqrt(ks * ks * c)
    dy = torch.randn(b, ho, wo, k, device=DEV, generator=gen)
    mask = torch.randn(x.shape, device=DEV, generator=gen)
    nchw = lambda t: t.double().permute(0, 3, 1, 2)                              # noqa: E731
    xr, wr = nchw(x).detach().requires_grad_(), nchw(wt).detach().requires_grad_()
    gx, gw = torch.autograd.grad(F.conv2d(xr, wr, None, stride=st, padding=p), (xr, wr), nchw(dy))
    want = dict(dgrad=torch.where(mask > 0, gx.permute(0, 2, 3, 1), torch.zeros((), dtype=torch.float64, device=DEV)),
                wgrad=gw.permute(0, 2, 3, 1), dbias=dy.double().sum((0, 1, 2)))
    errs = {}
    for mode, route in ((0, _lib.ROUTE_FP32), (6, _lib.ROUTE_SPLIT6), (9, _lib.ROUTE_SPLIT9)):
        g = _lib.with_route(geom, route)
        runs = []
        for _ in range(2):
            dx, dw, db = torch.full_like(x, float("nan")), torch.full_like(wt, float("nan")), torch.empty(k, device=DEV)
            folds, ws = _lib.FoldList(), _lib.conv_workspace(DEV)
            assert folds.conv2d_bwd_pair(dy, wt, mask, dx, x, dw, g, ws, dbias=db)
            folds.run()
            torch.cuda.synchronize()
            runs.append(dict(dgrad=dx, wgrad=dw, dbias=db))
        for op in want:
            assert torch.equal(runs[0][op], runs[1][op]), (mode, op)
            errs[mode, op] = _rel_rms(runs[0][op], want[op])
            assert errs[mode, op] <= 6e-7, (mode, op, errs[mode, op])
    for mode in (6, 9):
        for op in ("dgrad", "wgrad"):
            assert errs[mode, op] <= 1.5 * errs[0, op] + 3e-8, (mode, op, errs[mode, op], errs[0, op])


@pytest.mark.parametrize("mode", [9, 6])
def test_split_routes_on_tiny_operands(mode):
    """Operands of 2^-120 .. 2^-100: the third bf16 piece of such a number lies below 2^-126 -- a bf16 SUBNORMAL.
    Documented expectation: v_mfma_f32_32x32x16_bf16 takes subnormal bf16 inputs as they are (no flush), and the
    products here (>= 2^-240 in exact arithmetic) underflow fp32 anyway, so what this pins is (a) no NaN / Inf / garbage
    from subnormal pieces, (b) tiny x normal operands: x in 2^-120 .. 2^-100 against weights of magnitude 2^60 .. 2^90
    gives results of normal size that must carry the fp32 chain's accuracy (the pieces of x below 2^-126 contribute
    exactly or are lost at the 2^-24 level, never more).  Forward and weight gradient of conv 3's shape."""
    from accel_rl_amd import _lib
    b, h, w, c, k, ks, st, p = 8, 12, 9, 64, 64, 3, 1, 1
    gen = torch.Generator(device=DEV).manual_seed(29)
    geom = _lib.conv_geom(b, h, w, c, k, ks, ks, st, p, p, route=_lib.ROUTE_SPLIT9 if mode == 9 else _lib.ROUTE_SPLIT6)
    ws = _lib.conv_workspace(DEV)
    tiny = lambda shape: torch.randn(shape, device=DEV, generator=gen) * \
        torch.exp2(torch.randint(-120, -99, shape, device=DEV, generator=gen).float())     # noqa: E731
    big = lambda shape: torch.randn(shape, device=DEV, generator=gen) * \
        torch.exp2(torch.randint(60, 91, shape, device=DEV, generator=gen).float())        # noqa: E731
    x, wt = tiny((b, h, w, c)), big((k, ks, ks, c))
    assert (x != 0).all() and x.abs().max() < 2.0 ** -96
    y, yc = torch.empty(b, h, w, k, device=DEV), torch.empty(b, h, w, k, device=DEV)
    _lib.conv2d_fwd(x, wt, None, y, geom, False, ws)
    _lib.conv2d_fwd(x, wt, None, yc, _lib.with_route(geom, _lib.ROUTE_FP32), False, ws)       # the fp32 MFMA chain
    want = F.conv2d(x.double().permute(0, 3, 1, 2), wt.double().permute(0, 3, 1, 2), None, stride=1, padding=1).permute(0, 2, 3, 1)
    assert torch.isfinite(y).all()
    # operands spread over 2^20 x 2^30: the sums are carried by a few large products and every route rounds at the
    # size of its running sums -- the bar is the fp32 chain's own error on the same data, element by element against
    # the magnitude sum, and in rms
    scale = F.conv2d(x.double().abs().permute(0, 3, 1, 2), wt.double().abs().permute(0, 3, 1, 2), None, stride=1,
                     padding=1).permute(0, 2, 3, 1)
    e_split, e_chain = ((y.double() - want).abs() / scale).max().item(), ((yc.double() - want).abs() / scale).max().item()
    assert e_split <= max(2.0 * e_chain, 2.0 ** -22), (e_split, e_chain)
    assert _rel_rms(y, want) <= 1.5 * _rel_rms(yc, want) + 3e-8, (_rel_rms(y, want), _rel_rms(yc, want))
    # tiny x tiny: every product underflows; the result must be exactly +-0 or a denormal-size number, never garbage
    y2 = torch.full_like(y, float("nan"))
    _lib.conv2d_fwd(x, tiny((k, ks, ks, c)), None, y2, geom, False, ws)
    assert torch.isfinite(y2).all() and y2.abs().max() < 2.0 ** -126
    # weight gradient: dy big, x tiny
    dy = big((b, h, w, k))
    dw = torch.empty_like(wt)
    _lib.conv2d_bwd_weight(dy, x, dw, geom, ws)
    xr, wr = x.double().permute(0, 3, 1, 2).requires_grad_(), wt.double().permute(0, 3, 1, 2).requires_grad_()
    gw, = torch.autograd.grad(F.conv2d(xr, wr, None, stride=1, padding=1), wr, dy.double().permute(0, 3, 1, 2))
    dwc = torch.empty_like(wt)
    _lib.conv2d_bwd_weight(dy, x, dwc, _lib.with_route(geom, _lib.ROUTE_FP32), ws)
    assert torch.isfinite(dw).all()
    e_split, e_chain = _rel_rms(dw, gw.permute(0, 2, 3, 1)), _rel_rms(dwc, gw.permute(0, 2, 3, 1))
    assert e_split <= 1.5 * e_chain + 3e-8, (e_split, e_chain)


EXACT = [("conv2", 33, 25, 19, 32, 64, 4, 2, 1), ("conv3", 20, 12, 9, 64, 64, 3, 1, 1), ("dense", 512, 1, 1, 3456, 512, 1, 1, 0)]


@pytest.mark.parametrize("mode", [9, 6])
@pytest.mark.parametrize("layer", EXACT, ids=[c[0] for c in EXACT])
def test_split_routes_carry_full_fp32_significands_exactly(layer, mode):
    """The bf16-split routes must not lose a bit of an fp32 operand: x = h + m + l exactly, so when the OTHER operand of
    every multiply is a signed power of two (one bf16 piece) and every output is a single product, the result is the
    24-bit operand times that power of two -- exact in fp32, whatever order the piece products are accumulated in.
    One-hot filters (forward, data gradient) / one-hot dy (weight gradient) at the spec-1 shapes; reference = the same
    contraction in float64; bit-for-bit equality.  (A route that dropped or rounded pieces -- plain bf16 inputs, or a
    two-piece split -- fails this at the 2^-9 / 2^-17 level.)"""
    from accel_rl_amd import _lib
    lib = _lib.load()
    name, b, h, w, c, k, ks, st, p = layer
    gen = torch.Generator(device=DEV).manual_seed(11)
    geom = _lib.conv_geom(b, h, w, c, k, ks, ks, st, p, p)
    ho, wo = _lib.conv_out_hw(geom)
    ws = _lib.conv_workspace(DEV)
    pow2 = lambda shape: (torch.randint(0, 2, shape, device=DEV, generator=gen) * 2 - 1).float() * \
        torch.exp2(torch.randint(-3, 4, shape, device=DEV, generator=gen).float())          # noqa: E731
    full = lambda shape: torch.randn(shape, device=DEV, generator=gen) * 37.0                # noqa: E731  (24-bit significands)
    # one-hot filters: filter kk looks at ONE tap and ONE channel (channels distinct across the live filters, so that a
    # data-gradient element also receives a single product); filters beyond the channel count stay zero
    wt = torch.zeros(k, ks, ks, c, device=DEV)
    live = min(k, c)
    ty = torch.randint(0, ks, (live,), device=DEV, generator=gen)
    tx = torch.randint(0, ks, (live,), device=DEV, generator=gen)
    ch = torch.randperm(c, device=DEV, generator=gen)[:live]
    wt[torch.arange(live, device=DEV), ty, tx, ch] = pow2((live,))
    x, dy = full((b, h, w, c)), full((b, ho, wo, k))
    # one-hot dy for the weight gradient: channel kk is non-zero at ONE (image, pixel)
    dy1 = torch.zeros(b, ho, wo, k, device=DEV)
    dy1[torch.randint(0, b, (k,), device=DEV, generator=gen), torch.randint(0, ho, (k,), device=DEV, generator=gen),
        torch.randint(0, wo, (k,), device=DEV, generator=gen), torch.arange(k, device=DEV)] = pow2((k,))
    nchw = lambda t: t.double().permute(0, 3, 1, 2)                                          # noqa: E731
    xr, wr = nchw(x).detach().requires_grad_(), nchw(wt).detach().requires_grad_()
    out = F.conv2d(xr, wr, None, stride=st, padding=p)
    gx, = torch.autograd.grad(out, xr, nchw(dy), retain_graph=True)
    gw, = torch.autograd.grad(out, wr, nchw(dy1))
    want = dict(fwd=out.permute(0, 2, 3, 1).detach(), dgrad=gx.permute(0, 2, 3, 1), wgrad=gw.permute(0, 2, 3, 1))
    for t in want.values():
        assert torch.equal(t.float().double(), t)                                            # the references are fp32 numbers
    geom = _lib.with_route(geom, _lib.ROUTE_SPLIT9 if mode == 9 else _lib.ROUTE_SPLIT6)
    y, dx, dw = torch.empty(b, ho, wo, k, device=DEV), torch.empty_like(x), torch.empty_like(wt)
    _lib.conv2d_fwd(x, wt, None, y, geom, False, ws)
    _lib.conv2d_bwd_data(dy, wt, None, dx, geom)
    _lib.conv2d_bwd_weight(dy1, x, dw, geom, ws)
    torch.cuda.synchronize()
    for op, got in (("fwd", y), ("dgrad", dx), ("wgrad", dw)):
        assert want[op].abs().max() > 0
        assert torch.equal(got.double(), want[op]), (op, (got.double() - want[op]).abs().max().item())


def test_six_product_route_is_within_a_quarter_of_the_fp32_mfma_chains_error():
    """The accuracy gate behind bench.py's `alt_routes`: at every spec-1 layer and pass the
    six-product route's error against a float64 contraction of the same inputs stays within 1.25 x the fp32 MFMA
    chain's in rms (the reference's arithmetic is Theano floatX = float32: accel_rl/optimizers/single/ppo_optimizer.py:
    49-55), and within 1.5 x in the largest single error (one element out of 10^5 .. 10^7: a noisier statistic -- the
    dense forward's is 1.4 x on this seed for six AND for nine products).  The nine-product default is held to the same
    bars.  The table is the one bench.py prints as `accuracy`."""
    import bench
    table = bench.route_accuracy(DEV)
    assert set(table) == {"conv1", "conv2", "conv3", "dense1"}
    for layer, passes in table.items():
        for op, e in passes.items():
            rms, mx = e["rms_err_vs_f64"], e["max_err_vs_f64"]
            for route in ("split6", "split9"):
                assert rms[route] <= 1.25 * rms["fp32_mfma"], (layer, op, route, rms)
                assert mx[route] <= 1.5 * mx["fp32_mfma"], (layer, op, route, mx)
            assert rms["fp32_mfma"] < 1e-6 and mx["fp32_mfma"] < 3e-6, (layer, op, rms, mx)


@pytest.mark.parametrize("batch", [32, 64, 118, 119, 256])
@pytest.mark.parametrize("layer", [(25, 19, 32, 64, 4, 2, 1), (12, 9, 64, 64, 3, 1, 1)], ids=["conv2", "conv3"])
def test_tile_shapes_of_the_64_column_kernels_are_bit_identical(layer, batch):
    """The forward / data-gradient kernels of the 33 .. 64-column layers pick their tile by the launch's size (the column
    split up to 128 row tiles: batches 118 / 119 sit on either side of it): every shape issues the same piece products in
    the same order, so the choice can never show in a result (arl_dev_fwd_tile pins it for the comparison)."""
    from accel_rl_amd import _lib
    lib = _lib.load()
    h, w, c, k, ks, st, p = layer
    geom = _lib.conv_geom(batch, h, w, c, k, ks, ks, st, p, p)
    ho, wo = _lib.conv_out_hw(geom)
    ws = _lib.conv_workspace(DEV)
    gen = torch.Generator(device=DEV).manual_seed(batch)
    x = torch.randn(batch, h, w, c, device=DEV, generator=gen).relu()
    wt = torch.randn(k, ks, ks, c, device=DEV, generator=gen) / np.sqrt(ks * ks * c)
    bias = torch.randn(k, device=DEV, generator=gen)
    dy = torch.randn(batch, ho, wo, k, device=DEV, generator=gen)
    outs = []
    try:
        for v in (-1, 0, 1, 2):
            lib.arl_dev_fwd_tile(v)
            y, dx = torch.full((batch, ho, wo, k), float("nan"), device=DEV), torch.full_like(x, float("nan"))
            _lib.conv2d_fwd(x, wt, bias, y, geom, True, ws)
            _lib.conv2d_bwd_data(dy, wt, x, dx, geom)              # (masked by the layer's input, as the learner runs it)
            torch.cuda.synchronize()
            outs.append((y, dx))
    finally:
        lib.arl_dev_fwd_tile(-1)
    assert torch.isfinite(outs[0][0]).all() and torch.isfinite(outs[0][1]).all()
    for y, dx in outs[1:]:
        assert torch.equal(y, outs[0][0]) and torch.equal(dx, outs[0][1])


@pytest.mark.parametrize("case", [(512, 25, 19, 32, 64, 4, 2, 1),     # conv 2 of spec 1: stride 2, four parity classes
                                  (512, 12, 9, 64, 64, 3, 1, 1),      # conv 3 of spec 1 (128 x 64 tiles)
                                  (64, 12, 9, 64, 64, 3, 1, 1),       # ... at the column-split size
                                  (37, 25, 19, 32, 64, 4, 2, 1),      # ragged rows
                                  (96, 34, 26, 64, 64, 3, 1, 1),      # spec 2 / 3 style layers
                                  (48, 16, 12, 48, 128, 2, 2, 0),     # 2 x 2 stride 2: one tap per class, no padding
                                  (40, 12, 9, 20, 64, 3, 1, 1)])      # 20 columns
@pytest.mark.parametrize("precision", [9, 6])
def test_data_gradient_on_k_contiguous_weights_is_bit_identical(case, precision):
    """arl_conv2d_dgrad_weights + wt_or_null (ABI 4): the data gradient reads its own k-contiguous copy of the weights
    through the forward pass's loader -- same piece products, same order: dx bit for bit the one gathered from w, with
    and without the rectifier mask; the copy itself is the documented permutation of w."""
    from accel_rl_amd import _lib
    b, h, w, c, k, ks, st, p = case
    _lib.set_conv_precision(precision)
    try:
        x, wt, bias, geom, ws = _mk(case, seed=11)
        ho, wo = _lib.conv_out_hw(geom)
        dy = torch.randn(b, ho, wo, k, device=DEV, generator=torch.Generator(device=DEV).manual_seed(13))
        wk = torch.full_like(wt, float("nan")).reshape(-1)
        _lib.conv2d_dgrad_weights([(wt, wk, geom)])
        # the layout: class z = ph * st + pw, [c][(ty * taps + tx) * K + k] = w[k][i0 + st ty][j0 + st tx][c]
        taps = ks // st
        want = []
        for ph in range(st):
            for pw in range(st):
                i0, j0 = (ph + p) % st, (pw + p) % st
                sub = wt[:, i0::st, j0::st, :]                        # [K][taps][taps][C]
                want.append(sub.permute(3, 1, 2, 0).reshape(c, taps * taps * k))
        assert torch.equal(wk, torch.stack(want).reshape(-1))
        for mask in (None, x):
            _lib.load().arl_dev_dgrad_wt(0)
            dx0 = torch.full((b, h, w, c), float("nan"), device=DEV)
            _lib.conv2d_bwd_data(dy, wt, mask, dx0, geom, wt=wk)      # handed over, ignored: gathers from w
            dx1 = torch.full((b, h, w, c), float("nan"), device=DEV)
            _lib.conv2d_bwd_data(dy, wt, mask, dx1, geom)
            _lib.load().arl_dev_dgrad_wt(1)
            dx2 = torch.full((b, h, w, c), float("nan"), device=DEV)
            _lib.conv2d_bwd_data(dy, wt, mask, dx2, geom, wt=wk)
            assert torch.isfinite(dx2).all() and torch.equal(dx0, dx1) and torch.equal(dx2, dx1)
        # through the pair entry (separate launches for these shapes) with the deferred weight-gradient fold
        folds, ws2 = _lib.FoldList(), _lib.conv_workspace(DEV)
        dxp, dw = torch.empty_like(dx1), torch.empty_like(wt)
        folds.conv2d_bwd_pair(dy, wt, x, dxp, x, dw, geom, ws2, wt=wk)
        folds.run()
        assert torch.equal(dxp, dx2)
    finally:
        _lib.load().arl_dev_dgrad_wt(1)
        _lib.set_conv_precision(9)
