"""Convolution 1 read straight from the u8 observations (arl_conv2d_u8_fwd /
arl_conv2d_u8_bwd_weight_parts, csrc/mfma_conv.hip) against plain PyTorch fp32 on
float(obs[idx]) * scale, and against the two-kernel route it replaces
(arl_gather_scale_obs_nhwc + arl_conv2d_fwd / arl_conv2d_bwd_weight).
Floating point: |got - want| <= 2e-5 * sqrt(K_red) * max|want| (fp32 round-off of a
reduction in another order); run to run the kernels must be bit-identical."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SCALE = float(np.float32(1. / 255.))

#        rows  batch(idx)  C   H    W    K   kh  kw  stride
CASES = [(700, 512, 4, 104, 80, 32, 8, 8, 4),      # spec-1 conv 1 at the PPO minibatch, gathered rows
         (37, None, 4, 104, 80, 32, 8, 8, 4),      # ragged batch, rows in place
         (64, None, 4, 104, 80, 16, 8, 8, 4),      # spec-0 conv 1 (forward: the image kernel with half its tile idle; weight gradient: 16-wide tiles)
         (90, 33, 1, 104, 80, 16, 8, 8, 4),        # one frame per observation (the A2C example), ragged
         (21, 21, 3, 40, 36, 24, 4, 4, 4),         # 4-wide filter rows: four rows per k-tile
         (19, 7, 2, 48, 64, 8, 3, 16, 8),          # 16-wide filter rows: one row per k-tile, stride 8
         (3, 1, 4, 104, 80, 32, 8, 8, 4)]          # single image


@pytest.fixture(autouse=True, params=[9, 6, 0], ids=["split9", "split6", "fp32chain"])
def precision(request):
    """Every test under the three routes of arl_conv_geom.route (u8 pixels are exact in ONE bf16 piece: three piece
    products per k in both split routes; layers of <= 16 filters take the fp32 chain in every mode, except the 8 x 8 first layer's forward)."""
    from accel_rl_amd import _lib
    _lib.set_conv_precision(request.param)          # (the geometries built below take this module default)
    yield request.param
    _lib.set_conv_precision(9)


def _mk(case, seed=0):
    from accel_rl_amd import _lib
    rows, b, c, h, w, k, kh, kw, st = case
    gen = torch.Generator(device=DEV).manual_seed(seed)
    obs = torch.randint(0, 256, (rows, c, h, w), device=DEV, generator=gen, dtype=torch.int32).to(torch.uint8)
    idx = None
    if b is not None:
        idx = torch.randint(0, rows, (b,), device=DEV, generator=gen, dtype=torch.int32)
    wt = torch.randn(k, c, kh, kw, device=DEV, generator=gen) / np.sqrt(kh * kw * c)
    bias = torch.randn(k, device=DEV, generator=gen)
    geom = _lib.conv_geom(rows if b is None else b, h, w, c, k, kh, kw, st, 0, 0)
    return obs, idx, wt, bias, geom


def _x(obs, idx):
    rows = obs if idx is None else obs[idx.long()]
    return rows.float() * SCALE


def _tol(want, k_red):
    return 2e-5 * np.sqrt(k_red) * max(want.abs().max().item(), 1e-6)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("relu", [True, False])
def test_forward(case, relu):
    from accel_rl_amd import _lib
    obs, idx, wt, bias, geom = _mk(case)
    ho, wo = _lib.conv_out_hw(geom)
    y = torch.full((geom.batch, ho, wo, geom.out_c), float("nan"), device=DEV)
    _lib.conv2d_u8_fwd(obs, idx, SCALE, wt, bias, y, geom, relu)
    want = F.conv2d(_x(obs, idx), wt, bias, stride=case[8])
    if relu:
        want = F.relu(want)
    want = want.permute(0, 2, 3, 1)
    assert torch.isfinite(y).all()
    assert (y - want).abs().max().item() <= _tol(want, wt[0].numel())
    y2 = torch.empty_like(y)
    _lib.conv2d_u8_fwd(obs, idx, SCALE, wt, bias, y2, geom, relu)
    assert torch.equal(y, y2)


@pytest.mark.parametrize("case", CASES)
def test_weight_gradient(case):
    from accel_rl_amd import _lib
    obs, idx, wt, bias, geom = _mk(case, seed=1)
    ho, wo = _lib.conv_out_hw(geom)
    gen = torch.Generator(device=DEV).manual_seed(2)
    dy = torch.randn(geom.batch, ho, wo, geom.out_c, device=DEV, generator=gen)
    dy = torch.where(torch.rand(dy.shape, device=DEV, generator=gen) < 0.4, torch.zeros_like(dy), dy)
    outs = []
    for _ in range(2):
        dw = torch.full_like(wt, float("nan"))
        db = torch.full((geom.out_c,), float("nan"), device=DEV)
        folds, ws = _lib.FoldList(), _lib.conv_workspace(DEV)
        assert folds.conv2d_u8_bwd_weight(dy, obs, idx, SCALE, dw, geom, ws, dbias=db)
        folds.run()
        outs.append((dw, db))
    dw, db = outs[0]
    x = _x(obs, idx).requires_grad_(False)
    w_ref = wt.clone().requires_grad_(True)
    F.conv2d(x, w_ref, None, stride=case[8]).backward(dy.permute(0, 3, 1, 2))
    want = w_ref.grad
    k_red = geom.batch * ho * wo
    assert torch.isfinite(dw).all()
    assert (dw - want).abs().max().item() <= _tol(want, k_red)
    want_b = dy.sum(dim=(0, 1, 2))
    assert (db - want_b).abs().max().item() <= _tol(want_b, k_red)
    assert torch.equal(dw, outs[1][0]) and torch.equal(db, outs[1][1])


def test_matches_the_gather_route():
    """Against gather + scale followed by the NHWC kernels: the u8 kernels sum over the exact integer pixels plane
    by plane and apply the pixel scale to the finished sum, the other route scales every pixel first and sums pixel by
    pixel -- the same numbers up to f32 round-off."""
    from accel_rl_amd import _lib
    case = (300, 256, 4, 104, 80, 32, 8, 8, 4)
    obs, idx, wt, bias, geom = _mk(case, seed=3)
    ho, wo = _lib.conv_out_hw(geom)
    y = torch.empty(geom.batch, ho, wo, geom.out_c, device=DEV)
    _lib.conv2d_u8_fwd(obs, idx, SCALE, wt, bias, y, geom, True)
    x = torch.empty((geom.batch, 4, 104, 80), device=DEV, memory_format=torch.channels_last)
    _lib.gather_scale_obs_nhwc(obs, idx, x, SCALE)
    assert torch.equal(x, _x(obs, idx))                     # the loader's conversion is this arithmetic
    y2 = torch.empty_like(y)
    w_hwc = wt.permute(0, 2, 3, 1).contiguous()
    _lib.conv2d_fwd(x, w_hwc, bias, y2, geom, True, _lib.conv_workspace(DEV))
    assert (y - y2).abs().max().item() <= _tol(y2, 256)


def test_rejects_unsupported_geometry():
    from accel_rl_amd import _lib
    obs = torch.zeros(4, 4, 104, 80, dtype=torch.uint8, device=DEV)
    y = torch.empty(4, 25, 19, 64, device=DEV)
    wt = torch.empty(64, 4, 8, 8, device=DEV)
    geom = _lib.conv_geom(4, 104, 80, 4, 64, 8, 8, 4, 0, 0)           # 64 filters: not on this path
    assert not _lib.conv2d_u8_supported(104, 80, 64, 8, 8, 4, 0, 0)
    with pytest.raises(RuntimeError):
        _lib.conv2d_u8_fwd(obs, None, SCALE, wt, None, y, geom, True)


def _random_case(rs):
    kw = int(rs.choice([4, 8, 16]))
    kh = int(rs.choice([1, 2, 3, 4]) * (16 // kw))
    st = int(rs.choice([4, 8]))
    c = int(rs.randint(1, 6))
    k = int(rs.randint(1, 9) * 4)
    h = kh + st * int(rs.randint(0, 12)) + int(rs.randint(0, st))
    w = (kw + st * int(rs.randint(0, 12)) + 4 * int(rs.randint(0, 3)) + 3) // 4 * 4
    if (h * w) % 4:
        h += 1 if (w % 4 == 0) else 0
    rows = int(rs.randint(1, 70))
    b = None if rs.rand() < 0.4 else int(rs.randint(1, 90))
    return (rows, b, c, h, w, k, kh, kw, st)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("ARL_U8_RANDOM_CASES", "24"))))
def test_random_geometries(seed):
    """Random supported geometries (filter 1..4 k-tiles tall, 4 / 8 / 16 wide, stride 4 / 8, 1..5 planes,
    4..32 filters, ragged rows, gathered or in place): forward and weight gradient against PyTorch."""
    from accel_rl_amd import _lib
    rs = np.random.RandomState(1000 + seed)
    case = _random_case(rs)
    rows, b, c, h, w, k, kh, kw, st = case
    assert _lib.conv2d_u8_supported(h, w, k, kh, kw, st, 0, 0), case
    obs, idx, wt, bias, geom = _mk(case, seed=seed)
    ho, wo = _lib.conv_out_hw(geom)
    x = _x(obs, idx)
    y = torch.full((geom.batch, ho, wo, k), float("nan"), device=DEV)
    _lib.conv2d_u8_fwd(obs, idx, SCALE, wt, bias, y, geom, False)
    want = F.conv2d(x, wt, bias, stride=st).permute(0, 2, 3, 1)
    assert (y - want).abs().max().item() <= _tol(want, wt[0].numel()), case
    gen = torch.Generator(device=DEV).manual_seed(seed)
    dy = torch.randn(geom.batch, ho, wo, k, device=DEV, generator=gen)
    dw = torch.full_like(wt, float("nan"))
    db = torch.full((k,), float("nan"), device=DEV)
    folds = _lib.FoldList()
    assert folds.conv2d_u8_bwd_weight(dy, obs, idx, SCALE, dw, geom, _lib.conv_workspace(DEV), dbias=db)
    folds.run()
    w_ref = wt.clone().requires_grad_(True)
    F.conv2d(x, w_ref, None, stride=st).backward(dy.permute(0, 3, 1, 2))
    k_red = geom.batch * ho * wo
    assert (dw - w_ref.grad).abs().max().item() <= _tol(w_ref.grad, k_red), case
    want_b = dy.sum(dim=(0, 1, 2))
    assert (db - want_b).abs().max().item() <= _tol(want_b, k_red), case


IMG_CASES = [(600, 512, 4, 104, 80, 32, 8, 8, 4),      # the PPO minibatch: two images per workgroup, rows by index
             (256, None, 4, 104, 80, 32, 8, 8, 4),     # the rollout: one image per workgroup
             (300, 259, 4, 104, 80, 32, 8, 8, 4),      # three more images than CUs
             (40, 37, 1, 104, 80, 32, 8, 8, 4),        # one plane (four k-steps), fewer images than CUs
             (9, 7, 3, 48, 64, 32, 8, 8, 8),           # three planes, stride 8
             (5, None, 2, 200, 96, 32, 8, 8, 4)]       # a larger image (38 400 bytes, 1 128 output pixels: 36 row tiles)


@pytest.mark.parametrize("case", IMG_CASES)
@pytest.mark.parametrize("relu", [True, False])
def test_image_stationary_kernel_is_bit_identical_to_the_tap_gather_kernel(case, relu, precision):
    """csrc/img_conv.hip (32 filters of 8 x 8: the image in LDS, the weights split once per workgroup) against the kernel
    it replaces (igemm_body's U8 path, arl_dev_conv_variant(1)): the same piece products in the same order -- equal bit for
    bit on both split routes; on the fp32 chain both calls take the one old kernel.  And against PyTorch fp32."""
    from accel_rl_amd import _lib
    obs, idx, wt, bias, geom = _mk(case, seed=5)
    ho, wo = _lib.conv_out_hw(geom)
    outs = []
    for variant in (0, 1):
        _lib.load().arl_dev_conv_variant(variant)
        try:
            y = torch.full((geom.batch, ho, wo, geom.out_c), float("nan"), device=DEV)
            _lib.conv2d_u8_fwd(obs, idx, SCALE, wt, bias if relu else None, y, geom, relu)
            torch.cuda.synchronize()
        finally:
            _lib.load().arl_dev_conv_variant(0)
        outs.append(y)
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])
    want = F.conv2d(_x(obs, idx), wt, bias if relu else None, stride=case[8])
    want = (F.relu(want) if relu else want).permute(0, 2, 3, 1)
    assert (outs[0] - want).abs().max().item() <= _tol(want, wt[0].numel())
