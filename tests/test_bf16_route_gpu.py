"""ARL_CONV_ROUTE_BF16 (arl_conv_geom.route = 2): the labelled reduced-precision option of the contraction kernels -- fp32
operands rounded to nearest-even bf16 on their way into the matrix cores, ONE product per multiply, fp32 accumulation
("fp32 default; bf16 optional").  NOT a parity route: what is checked here is that the kernels
compute exactly what the label says -- the fp32 contraction of the ROUNDED operands (torch's `.bfloat16().float()` is
the same round-to-nearest-even), to the accumulation-order tolerance of the fp32 tests
(|got - want| <= 2e-5 sqrt(K_red) max|want|) -- that the result really is at bf16 distance from the fp32 routes
(so the route cannot silently be another one), that it is run-to-run bit-identical, and that a learner runs on it."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

#        batch  H    W   C   K  k  s  p
CASES = [(37, 104, 80, 4, 32, 8, 4, 0),      # spec-1 conv 1 from f32 rows (ragged batch)
         (32, 25, 19, 32, 64, 4, 2, 1),      # spec-1 conv 2 (stride-2 parity classes, padding)
         (16, 25, 19, 16, 32, 4, 2, 1),      # spec-0 conv 2
         (48, 12, 9, 64, 64, 3, 1, 1),       # spec-1 conv 3   (batches with whole 32-row k-tiles of output pixels: the
                                             #  weight gradient of a ragged one takes the generic fp32 kernels on every route)
         (512, 1, 1, 6912, 512, 1, 1, 0),    # spec-1 dense at the PPO minibatch (split-K forward)
         (256, 1, 1, 3840, 512, 1, 1, 0),    # odd k-tile count per split
         (5120, 1, 1, 512, 128, 1, 1, 0),    # 128 x 128 tiles
         (32, 1, 1, 256, 1152, 1, 1, 0)]     # C51 head at the DQN batch


def _r(t):
    return t.bfloat16().float()


def _tol(want, k_red):
    return 2e-5 * np.sqrt(k_red) * max(want.abs().max().item(), 1e-6)


def _mk(case, seed=0):
    from accel_rl_amd import _lib
    b, h, w, c, k, ks, st, p = case
    gen = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn(b, h, w, c, device=DEV, generator=gen)
    x = torch.where(torch.rand(x.shape, device=DEV, generator=gen) < 0.3, torch.zeros_like(x), x)
    wt = torch.randn(k, ks, ks, c, device=DEV, generator=gen) / np.sqrt(ks * ks * c)
    bias = torch.randn(k, device=DEV, generator=gen)
    geom = _lib.conv_geom(b, h, w, c, k, ks, ks, st, p, p, route=_lib.ROUTE_BF16)
    return x, wt, bias, geom, _lib.conv_workspace(DEV)


@pytest.mark.parametrize("case", CASES)
def test_forward_is_the_fp32_contraction_of_rounded_operands(case):
    from accel_rl_amd import _lib
    b, h, w, c, k, ks, st, p = case
    x, wt, bias, geom, ws = _mk(case)
    ho, wo = _lib.conv_out_hw(geom)
    y = torch.full((b, ho, wo, k), float("nan"), device=DEV)
    _lib.conv2d_fwd(x, wt, bias, y, geom, True, ws)
    want = F.relu(F.conv2d(_r(x).permute(0, 3, 1, 2), _r(wt).permute(0, 3, 1, 2), bias, stride=st, padding=p)).permute(0, 2, 3, 1)
    assert torch.isfinite(y).all()
    assert (y - want).abs().max().item() <= _tol(want, ks * ks * c), case
    y2 = torch.empty_like(y)
    _lib.conv2d_fwd(x, wt, bias, y2, geom, True, ws)
    assert torch.equal(y, y2)
    # ... and NOT the fp32 contraction of the operands themselves: at bf16 distance from the default route
    y9 = torch.empty_like(y)
    _lib.conv2d_fwd(x, wt, bias, y9, _lib.with_route(geom, _lib.ROUTE_SPLIT9), True, ws)
    rel = ((y - y9).pow(2).mean().sqrt() / y9.pow(2).mean().sqrt()).item()
    assert 2e-4 < rel < 2e-2, (case, rel)


@pytest.mark.parametrize("case", CASES)
def test_backward_is_the_fp32_contraction_of_rounded_operands(case):
    from accel_rl_amd import _lib
    b, h, w, c, k, ks, st, p = case
    x, wt, bias, geom, ws = _mk(case, seed=1)
    ho, wo = _lib.conv_out_hw(geom)
    dy = torch.randn(b, ho, wo, k, device=DEV, generator=torch.Generator(device=DEV).manual_seed(7))
    xr = _r(x).permute(0, 3, 1, 2).detach().requires_grad_()
    wr = _r(wt).permute(0, 3, 1, 2).detach().requires_grad_()
    gx, gw = torch.autograd.grad(F.conv2d(xr, wr, None, stride=st, padding=p), (xr, wr), _r(dy).permute(0, 3, 1, 2))
    gx, gw = gx.permute(0, 2, 3, 1), gw.permute(0, 2, 3, 1)
    if c <= 16:         # <= 16 output columns (here: the data gradient into <= 16 channels): the fp32 chain on EVERY route
        xf = x.permute(0, 3, 1, 2).detach().requires_grad_()
        gx = torch.autograd.grad(F.conv2d(xf, wt.permute(0, 3, 1, 2), None, stride=st, padding=p), xf,
                                 dy.permute(0, 3, 1, 2))[0].permute(0, 2, 3, 1)
    dx = torch.full((b, h, w, c), float("nan"), device=DEV)
    _lib.conv2d_bwd_data(dy, wt, None, dx, geom)
    assert (dx - gx).abs().max().item() <= _tol(gx, (ks // st) ** 2 * k), case
    dxm = torch.full((b, h, w, c), float("nan"), device=DEV)
    _lib.conv2d_bwd_data(dy, wt, x, dxm, geom)
    assert (dxm - gx * (x > 0)).abs().max().item() <= _tol(gx, (ks // st) ** 2 * k), case
    dw = torch.full_like(wt, float("nan"))
    _lib.conv2d_bwd_weight(dy, x, dw, geom, ws)
    assert (dw - gw).abs().max().item() <= _tol(gw, b * ho * wo), case
    # the layer's two gradients in one launch (what the learner calls) == the two separate calls, bit for bit
    folds, dx2, dw2 = _lib.FoldList(), torch.full_like(dx, float("nan")), torch.full_like(dw, float("nan"))
    folds.conv2d_bwd_pair(dy, wt, x, dx2, x, dw2, geom, ws)
    folds.run()
    assert torch.equal(dw2, dw), case
    # (a thin dense layer's data gradient splits its reduction in the paired call: tests/test_mfma_conv_gpu.py)
    split_ok = st == 1 and c > 64 and x.numel() <= 1 << 20
    assert torch.equal(dx2, dxm) or (split_ok and (dx2 - dxm).abs().max().item() <= _tol(gx, k)), case
    # a data gradient on a k-contiguous copy of the weights (atari_cnn_policy._dgrad_weight_items): same bits
    if 16 < c <= 64 and ks > 1:
        wtc = torch.empty_like(wt)
        _lib.conv2d_dgrad_weights([(wt, wtc, geom)])
        dx3 = torch.full_like(dx, float("nan"))
        _lib.conv2d_bwd_data(dy, wt, None, dx3, geom, wt=wtc)
        assert torch.equal(dx3, dx), case


@pytest.mark.parametrize("k", [32, 16])
def test_conv1_from_u8_rows(k):
    """conv 1 as the step runs it, straight from the planar u8 observations (the pixels are exact in bf16: only the
    weights -- forward -- and dy -- weight gradient -- are rounded); 16 filters take the fp32 chain on every route."""
    from accel_rl_amd import _lib
    b, c, h, w, ks, st = 40, 4, 104, 80, 8, 4
    gen = torch.Generator(device=DEV).manual_seed(5)
    obs = torch.randint(0, 256, (b, c, h, w), device=DEV, dtype=torch.int32, generator=gen).to(torch.uint8)
    w8 = torch.randn(k, c, ks, ks, device=DEV, generator=gen) / np.sqrt(ks * ks * c)
    bias = torch.randn(k, device=DEV, generator=gen)
    geom = _lib.conv_geom(b, h, w, c, k, ks, ks, st, 0, 0, route=_lib.ROUTE_BF16)
    ho, wo = _lib.conv_out_hw(geom)
    ws = _lib.conv_workspace(DEV)
    y = torch.full((b, ho, wo, k), float("nan"), device=DEV)
    _lib.conv2d_u8_fwd(obs, None, 1.0 / 255.0, w8, bias, y, geom, True)
    rw = _r(w8) if k > 16 else w8
    want = F.relu(F.conv2d(obs.float(), rw, None, stride=st) * (1.0 / 255.0) + bias.view(1, -1, 1, 1)).permute(0, 2, 3, 1)
    assert (y - want).abs().max().item() <= _tol(want, ks * ks * c)
    dy = torch.randn(b, ho, wo, k, device=DEV, generator=gen)
    rdy = _r(dy) if k > 16 else dy
    wr = w8.clone().requires_grad_()
    gw = torch.autograd.grad(F.conv2d(obs.float(), wr, None, stride=st), wr, rdy.permute(0, 3, 1, 2))[0] * (1.0 / 255.0)
    folds, dw, db = _lib.FoldList(), torch.full_like(w8, float("nan")), torch.full((k,), float("nan"), device=DEV)
    folds.conv2d_u8_bwd_weight(dy, obs, None, 1.0 / 255.0, dw, geom, ws, dbias=db)
    folds.run()
    assert (dw - gw).abs().max().item() <= _tol(gw, b * ho * wo)
    # the bias gradient is a column sum of dy itself: fp32 on every route
    assert (db - dy.sum((0, 1, 2))).abs().max().item() <= _tol(dy.sum((0, 1, 2)), b * ho * wo)


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "hipGraph"])
def test_ppo_learner_runs_on_the_bf16_route_and_stays_near_the_fp32_one(use_graph):
    """BASELINE config 2's learner (spec 1, 256 envs x 5, minibatch 512 x 4 epochs) from the same parameters on the same
    batch, once on the default route and once with every layer geometry stamped ARL_CONV_ROUTE_BF16
    (_lib.set_conv_precision(1) before the policy builds them): the behaviour policy's outputs agree to bf16 accuracy,
    the update is finite, the first minibatch's gradient norm (taken before any parameter moved) agrees within 2 % and
    the later ones (parameters already apart) within 25 %."""
    from accel_rl_amd import _lib
    from test_learner_gpu import make, fill
    n_env, horizon = 256, 5
    was = _lib.conv_precision()
    out = {}
    try:
        for mode in (9, 1):
            _lib.set_conv_precision(mode)
            policy, algo, buf, _ = make("ppo", n_env, horizon, use_graph, spec_id=1, n_act=4, minibatch=512, epochs=4)
            rs = np.random.RandomState(1)
            norms = []
            for itr in range(4 if use_graph else 1):
                fill(buf, policy, rs, n_env, horizon)
                if itr == 0:
                    prob = buf.agent_infos["prob"].clone()
                np.random.seed(5 + itr)
                _, infos = algo.optimize_policy(itr, buf)
                torch.cuda.synchronize()
                norms.append(infos["GradNorm"].clone())
            assert (algo._graph is not None) == use_graph
            out[mode] = (prob, norms[0], policy.flat_params.clone())
    finally:
        _lib.set_conv_precision(was)
    (p9, n9, f9), (p1, n1, f1) = out[9], out[1]
    assert torch.isfinite(f1).all() and torch.isfinite(n1).all()
    assert not torch.equal(p9, p1)                              # not the fp32 route under another name
    assert (p9 - p1).abs().max().item() < 2e-2
    assert abs((n9[0] - n1[0]).item()) <= 0.02 * n9[0].item(), (n9, n1)
    assert ((n9 - n1).abs() <= 0.25 * n9.abs() + 1e-3).all(), (n9, n1)
