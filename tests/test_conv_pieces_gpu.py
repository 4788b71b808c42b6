"""arl_conv_pieces: bf16 pieces of activations handed from the launch that produces a tensor to the launch that gathers
it next (csrc/mfma_conv.hip).  Everything here is exact: the pieces must sum to the fp32 tensor bit for bit, and a launch
that reads pieces must give the result of the launch that splits in the kernel bit for bit (same pieces, same products,
same order) -- the fp32 operand it no longer reads is poisoned with NaNs to prove it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sum_pieces(pc, like):
    """h + m + l in float64 (exact), shaped like the fp32 tensor."""
    return pc.double().sum(0).view(like.shape)


@pytest.fixture(autouse=True, params=[9, 6])
def precision(request):
    from accel_rl_amd import _lib
    assert _lib.load().arl_conv_precision(request.param) == 0
    yield request.param
    _lib.load().arl_conv_precision(9)


@pytest.mark.parametrize("b", [37, 256])
def test_forward_chain_on_pieces_is_bit_identical(b):
    """spec-1 trunk: conv 1 from u8 rows leaves pieces of y1, conv 2 reads them and leaves y2's, conv 3, dense."""
    from accel_rl_amd import _lib
    gen = torch.Generator(device=DEV).manual_seed(5)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=gen)                 # noqa: E731
    ws = _lib.conv_workspace(DEV)
    obs = torch.randint(0, 256, (b, 4, 104, 80), device=DEV, dtype=torch.int32, generator=gen).to(torch.uint8)
    g1 = _lib.conv_geom(b, 104, 80, 4, 32, 8, 8, 4, 0, 0)
    g2 = _lib.conv_geom(b, 25, 19, 32, 64, 4, 4, 2, 1, 1)
    g3 = _lib.conv_geom(b, 12, 9, 64, 64, 3, 3, 1, 1, 1)
    g4 = _lib.dense_geom(b, 6912, 512)
    assert _lib.conv_pieces_supported(g1, _lib.PIECES_U8FWD) == _lib.PIECES_OUT
    assert _lib.conv_pieces_supported(g2, _lib.PIECES_FWD) == _lib.PIECES_IN | _lib.PIECES_OUT
    assert _lib.conv_pieces_supported(g3, _lib.PIECES_FWD) == _lib.PIECES_IN | _lib.PIECES_OUT
    assert _lib.conv_pieces_supported(g4, _lib.PIECES_FWD) & _lib.PIECES_IN
    w1, b1 = rnd(32, 4, 8, 8) * 0.05, rnd(32)
    w2, b2 = rnd(64, 4, 4, 32) * 0.05, rnd(64)
    w3, b3 = rnd(64, 3, 3, 64) * 0.05, rnd(64)
    w4, b4 = rnd(512, 6912) * 0.02, rnd(512)
    y1, y2, y3, h = (torch.empty(b, 25, 19, 32, device=DEV), torch.empty(b, 12, 9, 64, device=DEV),
                     torch.empty(b, 12, 9, 64, device=DEV), torch.empty(b, 512, device=DEV))
    # reference: no pieces anywhere
    _lib.conv2d_u8_fwd(obs, None, 1. / 255, w1, b1, y1, g1, True)
    _lib.conv2d_fwd(y1, w2, b2, y2, g2, True, ws)
    _lib.conv2d_fwd(y2, w3, b3, y3, g3, True, ws)
    _lib.conv2d_fwd(y3, w4, b4, h, g4, True, ws)
    want = [t.clone() for t in (y1, y2, y3, h)]
    # with pieces; every fp32 operand that pieces replace is poisoned before its consumer runs
    p1, p2, p3 = _lib.pieces_like(y1), _lib.pieces_like(y2), _lib.pieces_like(y3)
    for t in (y1, y2, y3, h, p1, p2, p3):
        t.fill_(float("nan"))
    _lib.conv_pieces(None, p1)
    _lib.conv2d_u8_fwd(obs, None, 1. / 255, w1, b1, y1, g1, True)
    assert torch.equal(y1, want[0]) and torch.equal(_sum_pieces(p1, y1), y1.double())
    nan = torch.full_like(y1, float("nan"))
    _lib.conv_pieces(p1, p2)
    _lib.conv2d_fwd(nan, w2, b2, y2, g2, True, ws)
    assert torch.equal(y2, want[1]) and torch.equal(_sum_pieces(p2, y2), y2.double())
    nan = torch.full_like(y2, float("nan"))
    _lib.conv_pieces(p2, p3)
    _lib.conv2d_fwd(nan, w3, b3, y3, g3, True, ws)
    assert torch.equal(y3, want[2]) and torch.equal(_sum_pieces(p3, y3), y3.double())
    _lib.conv_pieces(p3, None)
    _lib.conv2d_fwd(torch.full_like(y3, float("nan")), w4, b4, h, g4, True, ws)
    assert torch.equal(h, want[3])
    # the pieces are consumed by the call that follows: the next call reads its fp32 operand again
    _lib.conv2d_fwd(want[1], w3, b3, y3, g3, True, ws)
    assert torch.equal(y3, want[2])


@pytest.mark.parametrize("b", [33, 512])
def test_backward_chain_on_pieces_is_bit_identical(b):
    """dense pair -> conv 3 pair (64-column data gradient) -> conv 2 pair (stride-2 parity classes into 32 channels):
    each data gradient leaves the pieces of its masked output for the next one."""
    from accel_rl_amd import _lib
    gen = torch.Generator(device=DEV).manual_seed(6)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=gen)                 # noqa: E731
    g2 = _lib.conv_geom(b, 25, 19, 32, 64, 4, 4, 2, 1, 1)
    g3 = _lib.conv_geom(b, 12, 9, 64, 64, 3, 3, 1, 1, 1)
    g4 = _lib.dense_geom(b, 6912, 512)
    for g in (g2, g3):
        assert _lib.conv_pieces_supported(g, _lib.PIECES_DGRAD) == _lib.PIECES_IN | _lib.PIECES_OUT
    # (the dense data gradient of a small batch splits its reduction: its output comes from the fold, without pieces)
    dense_out = bool(_lib.conv_pieces_supported(g4, _lib.PIECES_DGRAD) & _lib.PIECES_OUT)
    assert dense_out == (b == 512)
    w2, w3, w4 = rnd(64, 4, 4, 32) * 0.05, rnd(64, 3, 3, 64) * 0.05, rnd(512, 6912) * 0.02
    y1, y2, y3 = rnd(b, 25, 19, 32).relu(), rnd(b, 12, 9, 64).relu(), rnd(b, 12, 9, 64).relu()
    dh = rnd(b, 512)

    def run(pieces):
        folds = _lib.FoldList()
        wss = [_lib.conv_workspace(DEV) for _ in range(3)]
        d3, d2, d1 = torch.empty_like(y3), torch.empty_like(y2), torch.empty_like(y1)
        dw4, dw3, dw2 = torch.empty_like(w4), torch.empty_like(w3), torch.empty_like(w2)
        p3, p2 = (_lib.pieces_like(d3) if dense_out else None, _lib.pieces_like(d2)) if pieces else (None, None)
        for t in (d3, d2, d1, p3, p2):
            if t is not None:
                t.fill_(float("nan"))
        if p3 is not None:
            _lib.conv_pieces(None, p3)
        folds.conv2d_bwd_pair(dh, w4, y3.view(b, -1), d3.view(b, -1), y3.view(b, -1), dw4, g4, wss[0])
        if pieces:
            assert p3 is None or torch.equal(_sum_pieces(p3, d3), d3.double())
            _lib.conv_pieces(p3, p2)
        folds.conv2d_bwd_pair(d3, w3, y2, d2, y2, dw3, g3, wss[1])
        if pieces:
            assert torch.equal(_sum_pieces(p2, d2), d2.double())
            _lib.conv_pieces(p2, None)
        folds.conv2d_bwd_pair(d2, w2, y1, d1, y1, dw2, g2, wss[2])
        folds.run()
        torch.cuda.synchronize()
        return d3, d2, d1, dw4, dw3, dw2
    want, got = run(False), run(True)
    for a, c in zip(want, got):
        assert torch.isfinite(a).all() and torch.equal(a, c)


def test_routes_without_pieces_refuse_them():
    from accel_rl_amd import _lib
    lib = _lib.load()
    ws = _lib.conv_workspace(DEV)
    g = _lib.conv_geom(8, 25, 19, 16, 32, 4, 4, 2, 1, 1)            # 16 input channels: two taps per k-tile
    assert _lib.conv_pieces_supported(g, _lib.PIECES_FWD) == _lib.PIECES_OUT
    x, w, y = torch.randn(8, 25, 19, 16, device=DEV), torch.randn(32, 4, 4, 16, device=DEV), torch.empty(8, 12, 9, 32, device=DEV)
    _lib.conv_pieces(_lib.pieces_like(x), None)
    with pytest.raises(RuntimeError, match="pieces"):
        _lib.conv2d_fwd(x, w, None, y, g, False, ws)
    _lib.conv2d_fwd(x, w, None, y, g, False, ws)                    # ... and the refused call consumed them
    lib.arl_conv_precision(0)                                       # the fp32 MFMA chain has no pieces at all
    g3 = _lib.conv_geom(8, 12, 9, 64, 64, 3, 3, 1, 1, 1)
    assert _lib.conv_pieces_supported(g3, _lib.PIECES_FWD) == 0 and _lib.conv_pieces_supported(g3, _lib.PIECES_DGRAD) == 0
    x3, w3, y3 = torch.randn(8, 12, 9, 64, device=DEV), torch.randn(64, 3, 3, 64, device=DEV), torch.empty(8, 12, 9, 64, device=DEV)
    _lib.conv_pieces(None, _lib.pieces_like(y3))
    with pytest.raises(RuntimeError, match="pieces"):
        _lib.conv2d_fwd(x3, w3, None, y3, g3, False, ws)
    with pytest.raises(RuntimeError):
        _lib.conv_pieces_supported(_lib.conv_geom(0, 12, 9, 64, 64, 3, 3, 1, 1, 1), _lib.PIECES_FWD)


def test_policy_learner_step_is_bit_identical_with_and_without_pieces(monkeypatch):
    """AtariCnnPolicy hands pieces along the trunk when the routes take them: one PPO minibatch's loss and every
    gradient must equal the same step with pieces switched off (routes report no capability)."""
    from accel_rl_amd import _lib
    from accel_rl_amd.policies.atari_cnn_policy import AtariCnnPolicy
    from accel_rl_amd.policies.atari_cnn_specs import cnn_specs
    from accel_rl_amd.spaces import Discrete, UintBox, EnvSpec
    from accel_rl_amd.util.seed import set_seed
    real = _lib.conv_pieces_supported
    asked = []

    def step(off):
        monkeypatch.setattr(_lib, "conv_pieces_supported",
                            (lambda g, op: 0) if off else (lambda g, op: asked.append(real(g, op)) or asked[-1]))
        set_seed(3)
        pol = AtariCnnPolicy(**cnn_specs[1])
        pol.use_pieces = True
        pol.initialize(EnvSpec(UintBox((4, 104, 80)), Discrete(4)), device=DEV)
        gen = torch.Generator(device=DEV).manual_seed(9)
        n = 64
        obs = torch.randint(0, 256, (n, 4, 104, 80), device=DEV, dtype=torch.int32, generator=gen).to(torch.uint8)
        prob, value = pol.prob_value(obs)
        mb = dict(observations=obs, idx=None, actions=torch.randint(0, 4, (n,), device=DEV, dtype=torch.int32, generator=gen).to(torch.uint8),
                  advantages=torch.randn(n, device=DEV, generator=gen), returns=torch.randn(n, device=DEV, generator=gen),
                  old_prob=prob.clone() * 0.9 + 0.025, valids=None)
        loss4 = pol.loss_and_grads(mb, 1, 0.1, 1.0, 0.01, torch.full((1,), 0.7, device=DEV)).clone()
        torch.cuda.synchronize()
        return prob.clone(), value.clone(), pol.flat_grads.clone(), loss4
    a, c = step(False), step(True)
    assert any(v == _lib.PIECES_IN | _lib.PIECES_OUT for v in asked)
    for x, y in zip(a, c):
        assert torch.isfinite(x).all() and torch.equal(x, y)
    assert a[2].abs().max() > 0
