"""GPU parity tests: every HIP kernel, called through the C-ABI, against the
oracle (oracle/ref_port.py) and the committed golden vectors.

Bars: bit-exact for integer / byte / index outputs and for the scans (whose
dtype walk is reproduced exactly); 1e-5 relative for the standardisation (mean /
std are order-dependent reductions) and the optimiser step.
"""
import numpy as np
import pytest
import torch

import autograd_ref
from conftest import load_golden
from oracle import ref_port as P
from oracle import synth_ale

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def L():
    from accel_rl_amd import _lib
    _lib.load()
    return _lib


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def run_gae(L, r, v, d, lv, gam, lam, promo=0):
    n, t = r.shape
    adv = torch.empty(n * t, dtype=torch.float32, device=DEV)
    ret = torch.empty(n * t, dtype=torch.float32, device=DEV)
    L.gae_scan(dev(r.reshape(-1)), dev(v.reshape(-1)), dev(d.reshape(-1).astype(np.uint8)), dev(lv),
               gam, lam, n, t, adv, ret, promo=promo)
    return adv.cpu().numpy().reshape(n, t), ret.cpu().numpy().reshape(n, t)


def run_nstep(L, r, d, v, lv, gam, promo=0):
    n, t = r.shape
    adv = torch.empty(n * t, dtype=torch.float32, device=DEV)
    ret = torch.empty(n * t, dtype=torch.float32, device=DEV)
    L.nstep_return(dev(r.reshape(-1)), dev(d.reshape(-1).astype(np.uint8)), dev(v.reshape(-1)), dev(lv),
                   gam, n, t, ret, adv, promo=promo)
    return ret.cpu().numpy().reshape(n, t), adv.cpu().numpy().reshape(n, t)


# ----------------------------------------------------------------------------- scans

def test_gae_golden_bit_exact(L):
    g = load_golden("g1_g2_scans")
    for i in range(int(g["n_inputs"])):
        ik = "i%02d" % i
        for pi, (gam, lam) in enumerate(g["params"]):
            key = "%s_p%d" % (ik, pi)
            adv, ret = run_gae(L, g[ik + "_r"], g[ik + "_v"], g[ik + "_d"], g[ik + "_lv"], float(gam), float(lam))
            np.testing.assert_array_equal(adv, g[key + "_adv"], err_msg=key)
            np.testing.assert_array_equal(ret, g[key + "_ret"], err_msg=key)
            adv_l, _ = run_gae(L, g[ik + "_r"], g[ik + "_v"], g[ik + "_d"], g[ik + "_lv"], float(gam), float(lam), promo=1)
            np.testing.assert_array_equal(adv_l, g[key + "_adv_legacy"], err_msg=key + " legacy")


def test_nstep_golden_bit_exact(L):
    g = load_golden("g1_g2_scans")
    n = 0
    for i in range(int(g["n_inputs"])):
        ik = "i%02d" % i
        for pi, (gam, lam) in enumerate(g["params"]):
            key = "%s_p%d" % (ik, pi)
            if key + "_nret" not in g:
                continue
            ret, adv = run_nstep(L, g[ik + "_r"], g[ik + "_d"], g[ik + "_v"], g[ik + "_lv"], float(gam))
            np.testing.assert_array_equal(ret, g[key + "_nret"], err_msg=key)
            np.testing.assert_array_equal(adv, g[key + "_nadv"], err_msg=key)
            ret_l, _ = run_nstep(L, g[ik + "_r"], g[ik + "_d"], g[ik + "_v"], g[ik + "_lv"], float(gam), promo=1)
            np.testing.assert_array_equal(ret_l, g[key + "_nret_legacy"], err_msg=key + " legacy")
            n += 1
    assert n == 72


@pytest.mark.parametrize("n,t", [(256, 5), (1024, 5), (2048, 5), (1, 5), (257, 5), (1000, 1),
                                 (513, 2), (300, 8), (129, 9), (640, 16), (77, 17), (200, 32),
                                 (65, 34), (50, 35), (40, 128), (70, 136), (30, 137), (19, 544), (5, 545),
                                 (3, 1000)])
def test_scans_vs_oracle_shapes(L, n, t):
    """Ragged tiles, every kernel variant (LDS tiles of 256/128/64/32/8 envs, direct), both scans."""
    rs = np.random.RandomState(n * 1000 + t)
    r = rs.randn(n, t).astype(np.float32)
    v = (rs.randn(n, t) * 2).astype(np.float32)
    d = rs.rand(n, t) < 0.15
    lv = rs.randn(n).astype(np.float32)
    for promo, pname in ((0, "nep50"), (1, "legacy")):
        adv, ret = run_gae(L, r, v, d, lv, 0.99, 0.95, promo)
        a0, r0 = P.gae_scan(r, v, d, lv, 0.99, 0.95, pname)
        np.testing.assert_array_equal(adv, a0)
        np.testing.assert_array_equal(ret, r0)
        ret, adv = run_nstep(L, r, d, v, lv, 0.99, promo)
        r1, a1 = P.nstep_returns(r, d, v, lv, 0.99, pname)
        np.testing.assert_array_equal(ret, r1)
        np.testing.assert_array_equal(adv, a1)


@pytest.mark.parametrize("n,t", [(8, 128), (33, 32), (256, 5), (1000, 1), (513, 2), (300, 8), (129, 9), (77, 17),
                                 (65, 34), (50, 35), (200, 64), (40, 100), (31, 130), (70, 136), (30, 137), (9, 256),
                                 (19, 500), (7, 512), (5, 545)])
def test_wave_suffix_scan_within_tolerance(L, n, t):
    """ARL_PROMO_ASSOC: the scans as a wavefront suffix scan of affine maps (every steps-per-lane / lanes-per-segment
    shape: T <= 64 one step per lane, 2 / 4 / 8 steps per lane with vector accesses when T divides, scalar otherwise;
    T > 512 falls back to the exact walk).  Tolerance: BASELINE.json's 1e-5 for returns / advantages, relative to
    max(1, |x|), against BOTH exact modes of the oracle; identical results run to run."""
    L.load().arl_dev_scan_force_wave(1)                  # (by default horizons below 96 take the exact walk: it is faster there)
    try:
        ref = _wave_scan_checks(L, n, t)
        for groups in (1, 2, 4):                     # segment groups per wave (chosen by size otherwise): same bits
            L.load().arl_dev_scan_wave_groups(groups)
            got = _wave_scan_checks(L, n, t)
            for a, b in zip(ref, got):
                np.testing.assert_array_equal(a, b)
    finally:
        L.load().arl_dev_scan_force_wave(0)
        L.load().arl_dev_scan_wave_groups(0)


def _wave_scan_checks(L, n, t):
    rs = np.random.RandomState(n * 1000 + t)
    r = rs.randn(n, t).astype(np.float32)
    v = (rs.randn(n, t) * 2).astype(np.float32)
    d = rs.rand(n, t) < (0.15 if t < 64 else 0.03)
    lv = rs.randn(n).astype(np.float32)
    close = lambda got, want: np.all(np.abs(got - want) <= 1e-5 * np.maximum(1., np.abs(want)))     # noqa: E731
    adv, ret = run_gae(L, r, v, d, lv, 0.99, 0.95, promo=2)
    adv2, ret2 = run_gae(L, r, v, d, lv, 0.99, 0.95, promo=2)
    np.testing.assert_array_equal(adv, adv2)
    np.testing.assert_array_equal(ret, ret2)
    for pname in ("nep50", "legacy"):
        a0, r0 = P.gae_scan(r, v, d, lv, 0.99, 0.95, pname)
        assert close(adv, a0) and close(ret, r0), (pname, np.abs(adv - a0).max())
    a_leg, _ = P.gae_scan(r, v, d, lv, 0.99, 0.95, "legacy")
    assert np.mean(adv == a_leg) > 0.98              # reassociated f64 sums round to the same f32 almost everywhere
    nret, nadv = run_nstep(L, r, d, v, lv, 0.99, promo=2)
    for pname in ("nep50", "legacy"):
        r1, a1 = P.nstep_returns(r, d, v, lv, 0.99, pname)
        assert close(nret, r1) and close(nadv, a1), (pname, np.abs(nret - r1).max())
    # lambda = 1, gamma = 1, no terminals: advantages are plain suffix sums of the TD errors
    z = np.zeros_like(d)
    adv1, _ = run_gae(L, r, v, z, lv, 1.0, 1.0, promo=2)
    vn = np.concatenate([v[:, 1:], lv[:, None]], axis=1).astype(np.float64)
    want = np.cumsum((r.astype(np.float64) + vn - v)[:, ::-1], axis=1)[:, ::-1]
    assert np.all(np.abs(adv1 - want) <= 1e-5 * np.maximum(1., np.abs(want)))
    return adv, ret, nret, nadv, adv1


def test_scan_unaligned_views_take_the_direct_path(L):
    rs = np.random.RandomState(5)
    n, t = 100, 5
    r = rs.randn(n, t).astype(np.float32)
    v = rs.randn(n, t).astype(np.float32)
    d = rs.rand(n, t) < 0.2
    lv = rs.randn(n).astype(np.float32)
    pad = torch.zeros(n * t + 1, dtype=torch.float32, device=DEV)
    pad[1:] = dev(r.reshape(-1))
    adv = torch.empty(n * t, dtype=torch.float32, device=DEV)
    ret = torch.empty(n * t, dtype=torch.float32, device=DEV)
    L.gae_scan(pad[1:], dev(v.reshape(-1)), dev(d.reshape(-1).astype(np.uint8)), dev(lv), 0.99, 0.95, n, t, adv, ret)
    a0, r0 = P.gae_scan(r, v, d, lv, 0.99, 0.95)
    np.testing.assert_array_equal(adv.cpu().numpy().reshape(n, t), a0)
    np.testing.assert_array_equal(ret.cpu().numpy().reshape(n, t), r0)


def test_gae_full_size_properties(L):
    """BASELINE-scale and sweep-scale inputs: size-independent properties.
    (a) done everywhere => adv = r - v exactly; (b) lambda=0 => adv = delta_t;
    (c) linearity in rewards for done=0 (within fp32 rounding); (d) a random
    window re-checked against the oracle."""
    n, t = 1 << 20, 5
    gen = torch.Generator(device=DEV).manual_seed(3)
    r = torch.randn(n * t, device=DEV, generator=gen)
    v = torch.randn(n * t, device=DEV, generator=gen)
    lv = torch.randn(n, device=DEV, generator=gen)
    ones = torch.ones(n * t, dtype=torch.uint8, device=DEV)
    zeros = torch.zeros(n * t, dtype=torch.uint8, device=DEV)
    adv = torch.empty_like(r)
    ret = torch.empty_like(r)
    L.gae_scan(r, v, ones, lv, 0.99, 0.95, n, t, adv, ret)
    assert torch.equal(adv, (r.double() - v.double()).float())
    assert torch.equal(ret, adv + v)
    d = (torch.rand(n * t, device=DEV, generator=gen) < 0.1).to(torch.uint8)
    L.gae_scan(r, v, d, lv, 0.99, 0.95, n, t, adv, ret)
    lo = 12345
    a0, r0 = P.gae_scan(r.view(n, t)[lo:lo + 4096].cpu().numpy(), v.view(n, t)[lo:lo + 4096].cpu().numpy(),
                        d.view(n, t)[lo:lo + 4096].cpu().numpy(), lv[lo:lo + 4096].cpu().numpy(), 0.99, 0.95)
    np.testing.assert_array_equal(adv.view(n, t)[lo:lo + 4096].cpu().numpy(), a0)
    np.testing.assert_array_equal(ret.view(n, t)[lo:lo + 4096].cpu().numpy(), r0)
    # last tile too
    a1, _ = P.gae_scan(r.view(n, t)[-300:].cpu().numpy(), v.view(n, t)[-300:].cpu().numpy(),
                       d.view(n, t)[-300:].cpu().numpy(), lv[-300:].cpu().numpy(), 0.99, 0.95)
    np.testing.assert_array_equal(adv.view(n, t)[-300:].cpu().numpy(), a1)
    # linearity: scan(2r, 2v, 2lv) == 2 * scan(r, v, lv) exactly (powers of two commute with rounding)
    adv2 = torch.empty_like(r)
    ret2 = torch.empty_like(r)
    L.gae_scan(r * 2, v * 2, d, lv * 2, 0.99, 0.95, n, t, adv2, ret2)
    assert torch.equal(adv2, adv * 2)
    del zeros


# ----------------------------------------------------------------------------- valids / standardise

def test_valids_golden(L):
    g = load_golden("g3_valids")
    for c in range(int(g["n_cases"])):
        k = "c%02d" % c
        n, t = g[k + "_flags"].shape
        valids = torch.full((n * t,), 9, dtype=torch.int8, device=DEV)
        a, r, v = (dev(g[k + x].reshape(-1)) for x in ("_adv", "_ret", "_val"))
        L.valids_mask(dev(g[k + "_flags"].reshape(-1).astype(np.uint8)), n, t, valids, a, r, v)
        np.testing.assert_array_equal(valids.cpu().numpy().reshape(n, t), g[k + "_valids"], err_msg=k)
        np.testing.assert_array_equal(a.cpu().numpy().reshape(n, t), g[k + "_adv_z"])
        np.testing.assert_array_equal(r.cpu().numpy().reshape(n, t), g[k + "_ret_z"])
        np.testing.assert_array_equal(v.cpu().numpy().reshape(n, t), g[k + "_val_z"])


@pytest.mark.parametrize("n", [1, 5, 1280, 5120, 100003, 1 << 22])
@pytest.mark.parametrize("masked", [False, True])
def test_standardize_vs_oracle(L, n, masked):
    rs = np.random.RandomState(n % 9973)
    x = (rs.randn(n) * 3 + 1.5).astype(np.float32)
    valids = (rs.rand(n) < 0.7).astype(np.int8) if masked else None
    if masked:
        valids[0] = 1
    want = P.standardize(x, valids)
    xt = dev(x)
    ws = L.standardize_workspace(DEV)
    L.standardize(xt, None if valids is None else dev(valids), ws, 1e-6)
    got = xt.cpu().numpy()
    tol = 1e-5 * np.maximum(1.0, np.abs(want)) if n > 1 else 1e-5
    assert np.all(np.abs(got - want) <= tol), np.abs(got - want).max()
    if masked:
        np.testing.assert_array_equal(got[valids == 0], x[valids == 0])   # untouched


def test_process_samples_golden_pipeline(L):
    """scan -> valids -> standardise chained on the device == the reference's
    AdvActorCriticBase.process_samples (fixture G4)."""
    g = load_golden("g4_process_samples")
    ws = L.standardize_workspace(DEV)
    for c in range(int(g["n_cases"])):
        k = "c%02d" % c
        gam, lam, use_valids, std_adv = g[k + "_cfg"]
        n, t = g[k + "_r"].shape
        r, v, lv = dev(g[k + "_r"].reshape(-1)), dev(g[k + "_v"].reshape(-1)), dev(g[k + "_lv"])
        d = dev(g[k + "_d"].reshape(-1).astype(np.uint8))
        adv = torch.empty(n * t, dtype=torch.float32, device=DEV)
        ret = torch.empty(n * t, dtype=torch.float32, device=DEV)
        if lam == 1:
            L.nstep_return(r, d, v, lv, float(gam), n, t, ret, adv)
        else:
            L.gae_scan(r, v, d, lv, float(gam), float(lam), n, t, adv, ret)
        valids = None
        if use_valids:
            valids = torch.empty(n * t, dtype=torch.int8, device=DEV)
            L.valids_mask(dev(g[k + "_need"].reshape(-1).astype(np.uint8)), n, t, valids, adv, ret, v)
            np.testing.assert_array_equal(valids.cpu().numpy().reshape(n, t), g[k + "_valids"])
            np.testing.assert_array_equal(v.cpu().numpy().reshape(n, t), g[k + "_value_after"])
        np.testing.assert_array_equal(ret.cpu().numpy().reshape(n, t), g[k + "_ret"], err_msg=k)
        if std_adv:
            L.standardize(adv, valids, ws, 1e-6)
            want = g[k + "_adv"]
            got = adv.cpu().numpy().reshape(n, t)
            assert np.all(np.abs(got - want) <= 1e-5 * np.maximum(1, np.abs(want))), k
        else:
            np.testing.assert_array_equal(adv.cpu().numpy().reshape(n, t), g[k + "_adv"], err_msg=k)


# ----------------------------------------------------------------------------- sampling

def test_sampling_golden_bit_exact(L):
    g = load_golden("g5_sampling")
    for c in range(int(g["n_cases"])):
        k = "c%02d" % c
        p, u = g[k + "_prob"], g[k + "_u"]
        act = torch.full((len(u),), 255, dtype=torch.uint8, device=DEV)
        L.sample_categorical(dev(p), dev(u), act)
        np.testing.assert_array_equal(act.cpu().numpy(), g[k + "_act"], err_msg=k)


def test_sampling_large_vs_oracle(L):
    rs = np.random.RandomState(0)
    for b, a in ((1 << 18, 4), (100001, 18), (4099, 6)):
        p = rs.dirichlet(np.ones(a) * 0.3, size=b).astype(np.float32)
        u = rs.rand(b)
        u[:8] = [0.0, 1.0 - 1e-17, 0.999999999, 1e-300, 0.5, 0.25, 0.75, 0.9999]
        act = torch.empty(b, dtype=torch.uint8, device=DEV)
        L.sample_categorical(dev(p), dev(u), act)
        np.testing.assert_array_equal(act.cpu().numpy(), P.sample_actions(p, u))


# ----------------------------------------------------------------------------- pixels

def test_preprocess_vs_oracle(L):
    rs = np.random.RandomState(4)
    n = 37
    a = rs.randint(0, 256, size=(n, 210, 160), dtype=np.uint8)
    b = rs.randint(0, 256, size=(n, 210, 160), dtype=np.uint8)
    a[0] = 255; b[0] = 255          # saturation: (4*255+2)>>2 == 255
    a[1] = 0; b[1] = 0
    b[2] = a[2]
    out = torch.empty((n, 104, 80), dtype=torch.uint8, device=DEV)
    L.preprocess_frames(dev(a), dev(b), out)
    want = np.stack([P.preprocess_pair(a[i], b[i]) for i in range(n)])
    np.testing.assert_array_equal(out.cpu().numpy(), want)
    L.preprocess_frames(None, dev(b), out)
    want = np.stack([P.preprocess_pair(None, b[i]) for i in range(n)])
    np.testing.assert_array_equal(out.cpu().numpy(), want)


def test_preprocess_nearest_mode_vs_oracle(L):
    """ARL_RESAMPLE_NEAREST (SURVEY a-11 / 7.2: the mode atari_env.py:155 NAMES; box2x is what it computes and stays
    the default): dst(y, x) = max(a, b)(2y, 2x), bit for bit, with and without a first frame; and the two modes differ
    on random frames (the switch is not a no-op)."""
    rs = np.random.RandomState(14)
    n = 19
    a = rs.randint(0, 256, size=(n, 210, 160), dtype=np.uint8)
    b = rs.randint(0, 256, size=(n, 210, 160), dtype=np.uint8)
    out = torch.empty((n, 104, 80), dtype=torch.uint8, device=DEV)
    L.preprocess_frames(dev(a), dev(b), out, resample="nearest")
    want = np.stack([P.preprocess_pair(a[i], b[i], "nearest") for i in range(n)])
    np.testing.assert_array_equal(out.cpu().numpy(), want)
    np.testing.assert_array_equal(want, np.maximum(a, b)[:, 0:208:2, 0::2])
    L.preprocess_frames(None, dev(b), out, resample="nearest")
    np.testing.assert_array_equal(out.cpu().numpy(), b[:, 0:208:2, 0::2])
    box = torch.empty_like(out)
    L.preprocess_frames(None, dev(b), box)
    assert (box != out).float().mean().item() > 0.9


def test_preprocess_bank_frames(L):
    bank = synth_ale.frame_bank(1)
    out = torch.empty((64, 104, 80), dtype=torch.uint8, device=DEV)
    L.preprocess_frames(dev(bank), dev(np.roll(bank, 1, axis=0)), out)
    want = np.stack([P.preprocess_pair(bank[i], bank[i - 1]) for i in range(64)])
    np.testing.assert_array_equal(out.cpu().numpy(), want)


def test_gather_scale(L):
    rs = np.random.RandomState(8)
    obs = rs.randint(0, 256, size=(1280, 4, 104, 80), dtype=np.uint8)
    idx = rs.permutation(1280)[:512].astype(np.int32)
    out = torch.empty((512, 4, 104, 80), dtype=torch.float32, device=DEV)
    L.gather_scale_obs(dev(obs), dev(idx), out, 1. / 255)
    want = obs[idx].astype(np.float32) * np.float32(1. / 255)
    np.testing.assert_array_equal(out.cpu().numpy(), want)
    out2 = torch.empty((1280, 4, 104, 80), dtype=torch.float32, device=DEV)
    L.gather_scale_obs(dev(obs), None, out2, 1. / 255)
    np.testing.assert_array_equal(out2.cpu().numpy(), obs.astype(np.float32) * np.float32(1. / 255))


# ----------------------------------------------------------------------------- optimiser

def _opt_state(L, p, g, with_v=True, log_len=16):
    n = p.numel()
    st = L.ArlOptState()
    keep = dict(p=p, g=g, m=torch.zeros_like(p), v=torch.zeros_like(p) if with_v else None,
                t=torch.zeros(1, device=DEV), lr=torch.ones(1, device=DEV),
                part=torch.zeros(L.OPT_PARTIALS, dtype=torch.float64, device=DEV),
                log=torch.zeros(log_len, device=DEV))
    st.n_params = n
    st.params, st.grads, st.slot0 = p.data_ptr(), g.data_ptr(), keep["m"].data_ptr()
    st.slot1 = keep["v"].data_ptr() if with_v else None
    st.step_count, st.lr_mult = keep["t"].data_ptr(), keep["lr"].data_ptr()
    st.partials, st.grad_norm_log, st.norm_log_len = keep["part"].data_ptr(), keep["log"].data_ptr(), log_len
    return st, keep


@pytest.mark.parametrize("n", [3, 1000, 898613, 3620005])
@pytest.mark.parametrize("clip", [None, 0.5])
def test_adam_vs_oracle(L, n, clip):
    rs = np.random.RandomState(n % 1000)
    p0 = rs.randn(n).astype(np.float32)
    p, g = dev(p0), torch.zeros(n, device=DEV)
    st, keep = _opt_state(L, p, g)
    m = np.zeros(n, np.float32); v = np.zeros(n, np.float32); t = np.float32(0); pw = p0.copy()
    for it in range(3):
        gh = (rs.randn(n) * (0.01 if it else 3.0)).astype(np.float32)
        g.copy_(dev(gh))
        keep["lr"].fill_(1.0 - 0.25 * it)
        L.opt_step(st, L.OPT_ADAM, 1e-3, 0.5, clip, 0.9, 0.999, 1e-5)
        gavg = (gh * np.float32(0.5)).astype(np.float32)
        gc, norm = P.clip_by_total_norm(gavg, clip)
        pw, m, v, t = P.adam_step(pw, gc, m, v, t, np.float32(1e-3) * np.float32(1.0 - 0.25 * it), eps=1e-5)
        got = p.cpu().numpy()
        assert np.allclose(got, pw, rtol=1e-5, atol=1e-6), (it, np.abs(got - pw).max())
        assert abs(keep["log"][it].item() - norm) <= 1e-5 * max(1, norm)
    assert keep["t"].item() == 3.0


@pytest.mark.parametrize("method", ["adam", "rmsprop"])
@pytest.mark.parametrize("n", [3, 898613, 3620005])
def test_noclip_single_launch_update_vs_oracle(L, n, method):
    """arl_opt_step_noclip / arl_opt_finish: calls of 3, 2 and 5 updates (odd and even lengths: Lasagne's t ping-pongs
    between two words) -- parameters after every update, the call's logged norms and t against the oracle."""
    rs = np.random.RandomState(n % 997)
    p0 = rs.randn(n).astype(np.float32)
    p, g = dev(p0), torch.zeros(n, device=DEV)
    adam = method == "adam"
    st, keep = _opt_state(L, p, g, with_v=adam)
    step_pp = torch.zeros(2, device=DEV)
    parts = torch.zeros(L.OPT_NORM_SLOTS * L.OPT_NORM_BLOCKS, dtype=torch.float64, device=DEV)
    m = np.zeros(n, np.float32); v = np.zeros(n, np.float32); t = np.float32(0); pw = p0.copy()
    for call, n_upd in enumerate((3, 2, 5)):
        norms = []
        for k in range(n_upd):
            gh = (rs.randn(n) * (0.01 if (call + k) else 3.0)).astype(np.float32)
            g.copy_(dev(gh))
            keep["lr"].fill_(1.0 - 0.1 * k)
            lr = np.float32(1e-3 if adam else 7e-4) * np.float32(1.0 - 0.1 * k)
            gavg = (gh * np.float32(0.5)).astype(np.float32)
            _, norm = P.clip_by_total_norm(gavg, None)
            norms.append(norm)
            if adam:
                L.opt_step_noclip(st, L.OPT_ADAM, 1e-3, 0.5, 0.9, 0.999, 1e-5, k, step_pp, parts)
                pw, m, v, t = P.adam_step(pw, gavg, m, v, t, lr, eps=1e-5)
            else:
                L.opt_step_noclip(st, L.OPT_RMSPROP, 7e-4, 0.5, 0.9, 0.0, 1e-6, k, step_pp, parts)
                pw, m = P.rmsprop_step(pw, gavg, m, lr)
                t = np.float32(t + 1)
            got = p.cpu().numpy()
            assert np.allclose(got, pw, rtol=1e-5, atol=1e-6), (call, k, np.abs(got - pw).max())
            assert keep["t"].item() == float(t)
        L.opt_finish(st, n_upd, 0.5, step_pp, parts)
        log = keep["log"].cpu().numpy()[:n_upd]
        assert np.allclose(log, norms, rtol=1e-5), (call, log, norms)
        assert keep["t"].item() == float(t) and step_pp.cpu().tolist() == [float(t)] * 2


@pytest.mark.parametrize("method", ["adam", "rmsprop"])
@pytest.mark.parametrize("carrier", ["own launch", "data gradient"])
def test_split_update_is_the_plain_update(L, method, carrier):
    """arl_opt_step_noclip_split / arl_corun_job: the update of a hole of the bucket as its own launch, or in
    extra workgroups of a data-gradient launch (whose own result must not change), plus the update of the rest ==
    the one-launch update bit for bit: parameters, slots, t, and the call's logged norms (same f64 partial sums up to
    their grouping)."""
    n, first, count = 3620004, 57312, 3538944                   # spec 1's bucket and its first dense weight tensor
    adam = method == "adam"
    mid, args = (L.OPT_ADAM, (1e-3, 0.5, 0.9, 0.999, 1e-5)) if adam else (L.OPT_RMSPROP, (7e-4, 0.5, 0.9, 0.0, 1e-6))
    rs = np.random.RandomState(3)
    p0 = rs.randn(n).astype(np.float32)
    twins = []
    for _ in range(2):
        p, g = dev(p0), torch.zeros(n, device=DEV)
        st, keep = _opt_state(L, p, g, with_v=adam)
        twins.append((st, keep, p, g, torch.zeros(2, device=DEV),
                      torch.zeros(L.OPT_NORM_SLOTS * L.OPT_NORM_BLOCKS, dtype=torch.float64, device=DEV)))
    # the hosting launch: conv 3's data gradient at 64 images
    geom = L.conv_geom(64, 12, 9, 64, 64, 3, 3, 1, 1, 1)
    gen = torch.Generator(device=DEV).manual_seed(5)
    dy = torch.randn(64, 12, 9, 64, device=DEV, generator=gen)
    wt = torch.randn(64, 3, 3, 64, device=DEV, generator=gen) * 0.05
    dx_ref = torch.empty(64, 12, 9, 64, device=DEV)
    L.conv2d_bwd_data(dy, wt, None, dx_ref, geom)
    for call, n_upd in enumerate((3, 2)):
        for k in range(n_upd):
            gh = (rs.randn(n) * (0.01 if (call + k) else 3.0)).astype(np.float32)
            for st, keep, p, g, step_pp, parts in twins:
                g.copy_(dev(gh))
                keep["lr"].fill_(1.0 - 0.1 * k)
            st, keep, p, g, step_pp, parts = twins[0]
            L.opt_step_noclip(st, mid, *args, k, step_pp, parts)
            st, keep, p, g, step_pp, parts = twins[1]
            if carrier == "own launch":
                L.opt_step_noclip_split(st, mid, *args, k, step_pp, parts, first, count, 1)
            else:
                job = L.corun_job(st, mid, *args, k, step_pp, parts, first, count)
                dx = torch.full_like(dx_ref, float("nan"))
                assert L.conv2d_bwd_data(dy, wt, None, dx, geom, corun=job)     # the launch took the job
                assert torch.equal(dx, dx_ref)
            L.opt_step_noclip_split(st, mid, *args, k, step_pp, parts, first, count, 0)
            for key in ("p", "m") + (("v",) if adam else ()):
                assert torch.equal(twins[0][1][key], twins[1][1][key]), (call, k, key)
            assert twins[0][1]["t"].item() == twins[1][1]["t"].item()
        L.opt_finish(twins[0][0], n_upd, 0.5, twins[0][4], twins[0][5])
        L.opt_finish(twins[1][0], n_upd, 0.5, twins[1][4], twins[1][5], hole_count=count)
        a, b = twins[0][1]["log"][:n_upd].cpu().numpy(), twins[1][1]["log"][:n_upd].cpu().numpy()
        assert np.allclose(a, b, rtol=1e-6), (a, b)
        assert twins[0][4].cpu().tolist() == twins[1][4].cpu().tolist()
    # a launch that cannot carry a job says so (channel counts 12 / 20: the generic kernels) and leaves the bucket
    # alone; the job then runs on its own; a job that is never run costs nothing (plain data, nothing pending)
    st, keep, p, g, step_pp, parts = twins[1]
    before = keep["p"].clone()
    job = L.corun_job(st, mid, *args, 0, step_pp, parts, first, count)
    ogeom = L.conv_geom(9, 20, 14, 12, 20, 3, 3, 1, 1, 1)
    odx = torch.empty(9, 20, 14, 12, device=DEV)
    assert not L.conv2d_bwd_data(torch.randn(9, 20, 14, 20, device=DEV), torch.randn(20, 3, 3, 12, device=DEV), None, odx,
                                 ogeom, corun=job)
    torch.cuda.synchronize()
    assert torch.equal(before, keep["p"])
    L.corun_job_run(job)
    torch.cuda.synchronize()
    assert not torch.equal(before[first:first + count], keep["p"][first:first + count])
    assert torch.equal(before[:first], keep["p"][:first]) and torch.equal(before[first + count:], keep["p"][first + count:])


@pytest.mark.parametrize("clip", [None, 0.5])
def test_rmsprop_vs_oracle(L, clip):
    n = 898613
    rs = np.random.RandomState(2)
    p0 = rs.randn(n).astype(np.float32)
    p, g = dev(p0), torch.zeros(n, device=DEV)
    st, keep = _opt_state(L, p, g, with_v=False)
    acc = np.zeros(n, np.float32); pw = p0.copy()
    for it in range(3):
        gh = rs.randn(n).astype(np.float32)
        g.copy_(dev(gh))
        L.opt_step(st, L.OPT_RMSPROP, 7e-4, 1.0, clip, 0.9, 0.0, 1e-6)
        gc, norm = P.clip_by_total_norm(gh, clip)
        pw, acc = P.rmsprop_step(pw, gc, acc, 7e-4)
        got = p.cpu().numpy()
        assert np.allclose(got, pw, rtol=1e-5, atol=1e-6), (it, np.abs(got - pw).max())


# ----------------------------------------------------------------------------- learner glue

def test_gather_scale_nhwc(L):
    rs = np.random.RandomState(9)
    obs = rs.randint(0, 256, size=(300, 4, 104, 80), dtype=np.uint8)
    idx = rs.permutation(300)[:77].astype(np.int32)
    out = torch.empty((77, 104, 80, 4), dtype=torch.float32, device=DEV)
    L.gather_scale_obs_nhwc(dev(obs), dev(idx), out, 1. / 255)
    want = obs[idx].transpose(0, 2, 3, 1).astype(np.float32) * np.float32(1. / 255)
    np.testing.assert_array_equal(out.cpu().numpy(), want)
    out2 = torch.empty((300, 104, 80, 4), dtype=torch.float32, device=DEV)
    L.gather_scale_obs_nhwc(dev(obs), None, out2, 1. / 255)
    np.testing.assert_array_equal(out2.cpu().numpy(), obs.transpose(0, 2, 3, 1).astype(np.float32) * np.float32(1. / 255))


@pytest.mark.parametrize("rows,ch", [(512 * 475, 32), (512 * 108, 64), (512, 512), (7, 4), (1000, 256), (3, 1024)])
def test_bias_relu_and_backward(L, rows, ch):
    rs = np.random.RandomState(rows % 97)
    x = rs.randn(rows, ch).astype(np.float32)
    b = rs.randn(ch).astype(np.float32)
    xt, bt = dev(x), dev(b)
    L.bias_relu(xt, bt, rows, ch)
    y = np.maximum(x + b, np.float32(0))
    np.testing.assert_array_equal(xt.cpu().numpy(), y)
    dy = rs.randn(rows, ch).astype(np.float32)
    dyt, db = dev(dy), torch.full((ch,), 7.0, device=DEV)
    ws = L.relu_bwd_workspace(DEV)
    L.relu_bwd_bias_grad(dyt, xt, rows, ch, db, ws)
    want = dy * (y > 0)
    np.testing.assert_array_equal(dyt.cpu().numpy(), want)
    ref = want.astype(np.float64).sum(axis=0)
    assert np.allclose(db.cpu().numpy(), ref, rtol=1e-4, atol=1e-3 * np.sqrt(rows))
    db2 = torch.zeros(ch, device=DEV)                     # deterministic: bit-identical on a re-run
    dyt2 = dev(dy)
    L.relu_bwd_bias_grad(dyt2, xt, rows, ch, db2, ws)
    assert torch.equal(db, db2)


@pytest.mark.parametrize("n_act,hid,batch", [(4, 512, 512), (6, 256, 100), (18, 512, 64), (4, 64, 5), (9, 1024, 33)])
@pytest.mark.parametrize("kind,tie", [(0, "theano"), (1, "theano"), (1, "math"), (1, "both")])
@pytest.mark.parametrize("masked", [False, True])
def test_pg_head_loss_vs_autograd(L, n_act, hid, batch, kind, tie, masked):
    """heads + softmax + A2C/PPO/value/entropy losses and all gradients vs PyTorch autograd
    on the reference's formulas (aac_base.py:60-70, a2c.py:43-46, ppo.py:42-51); PPO under the three gradient rules
    for the surrogate's min / clip: the reference's Theano (>= 0.8: a tie goes to the first argument, the default), the
    mathematical derivative, and Theano <= 0.7 (a tie feeds both arguments: 2 A inside the clip range)."""
    gen = torch.Generator(device=DEV).manual_seed(n_act * 1000 + hid + batch)
    n_rows = batch * 3
    h = torch.relu(torch.randn(batch, hid, device=DEV, generator=gen)).requires_grad_()
    w = (torch.randn(n_act + 1, hid, device=DEV, generator=gen) * 0.05).requires_grad_()
    bh = (torch.randn(n_act + 1, device=DEV, generator=gen) * 0.1).requires_grad_()
    act = torch.randint(0, n_act, (n_rows,), device=DEV, generator=gen).to(torch.uint8)
    adv = torch.randn(n_rows, device=DEV, generator=gen)
    ret = torch.randn(n_rows, device=DEV, generator=gen)
    old = torch.softmax(torch.randn(n_rows, n_act, device=DEV, generator=gen), 1)
    idx = torch.randperm(n_rows, device=DEV, generator=gen)[:batch].to(torch.int32)
    valids = (torch.rand(n_rows, device=DEV, generator=gen) < 0.7).to(torch.int8) if masked else None
    lr_mult = torch.full((1,), 0.6, device=DEV)
    clip, c_v, c_e = 0.2, 0.25 if kind == 0 else 1.0, 0.01
    sel = idx.long()
    if masked:
        valids[sel[0]] = 1
        inv = (1. / valids[sel].sum(dtype=torch.float32)).reshape(1)
        wgt = valids[sel].float() * inv
    else:
        inv, wgt = None, torch.full((batch,), 1. / batch, device=DEV)
    # --- autograd reference
    out = h @ w.t() + bh
    prob, value = torch.softmax(out[:, :n_act], 1), out[:, n_act]
    a = act[sel].long()
    pa = prob[torch.arange(batch), a]
    if kind == 1:
        # make a good share of the ratios leave the clip range
        old_sel = old[sel]
        ratio = (pa + 1e-8) / (old_sel[torch.arange(batch), a] + 1e-8)
        c = clip * 0.6
        pi = -torch.sum(wgt * autograd_ref.ppo_surrogate(ratio, adv[sel], c, tie))
    else:
        pi = -torch.sum(wgt * torch.log(pa + 1e-8) * adv[sel])
    vl = c_v * torch.sum(wgt * (value - ret[sel]) ** 2)
    el = -c_e * torch.sum(wgt * -torch.sum(prob * torch.log(prob + 1e-8), dim=1))
    gh, gw, gb = torch.autograd.grad(pi + vl + el, [h, w, bh])
    # --- kernel
    dout = torch.empty(batch, n_act + 1, device=DEV)
    dh = torch.empty(batch, hid, device=DEV)
    dw, db, loss4 = torch.empty_like(w), torch.empty_like(bh), torch.zeros(4, device=DEV)
    ws = L.pg_head_workspace(DEV)
    L.pg_head_loss(h.detach(), w.detach(), bh.detach(), act, adv, ret, old, valids, idx, lr_mult, inv,
                   n_act, kind, clip, c_v, c_e, dout, dh, dw, db, loss4, ws,
                   tie_rule=dict(theano=L.PPO_TIE_THEANO, math=L.PPO_TIE_MATH, both=L.PPO_TIE_BOTH)[tie])
    assert torch.allclose(loss4[:3], torch.stack([pi, vl, el]).detach(), rtol=1e-4, atol=1e-6)
    for got, want, name in ((dh, gh, "dh"), (dw, gw, "dw"), (db, gb, "db")):
        scale = max(want.abs().max().item(), 1e-6)
        assert torch.allclose(got, want, rtol=1e-3, atol=1e-5 * scale), (name, (got - want).abs().max().item(), scale)
    # --- inference twin
    p2 = torch.empty(batch, n_act, device=DEV)
    v2 = torch.empty(batch, device=DEV)
    L.pg_head_infer(h.detach(), w.detach(), bh.detach(), p2, v2)
    assert torch.allclose(p2, prob.detach(), rtol=1e-5, atol=1e-7) and torch.allclose(v2, value.detach(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tie", ["theano", "both", "math"])
def test_ppo_tie_rule_on_the_boundaries_at_the_config2_shape(L, tie):
    """BASELINE config 2's minibatch (512 rows, 4 actions, 512 hidden units).  The samples the two gradient rules
    differ on, placed deliberately: ratio == 1 exactly (old probability = the kernel's own), ratio exactly ON the
    lower / upper bound (clip chosen as 1 - ratio, resp. ratio - 1, of a row: both exact in fp32), advantage == 0,
    clip == 0 (both bounds coincide with ratio 1), and ordinary rows either side of the range.  The reference's rule
    (ppo.py:47-49 through Theano >= 0.8's Minimum.L_op / Clip.L_op): A wherever the unclipped branch IS the minimum --
    inside the range, bounds INCLUDED, it ties with the clipped one and takes the whole gradient --, 0 otherwise;
    "both" (Theano <= 0.7): 2 A inside the range, bounds included.  The reference side is autograd on the same formulas with
    the forward value of the ratio pinned to the kernel's own bits (straight-through), so that a last-bit difference
    between torch's softmax and the kernel's cannot move a sample across a bound."""
    batch, n_act, hid = 512, 4, 512
    gen = torch.Generator(device=DEV).manual_seed(77)
    h0 = torch.relu(torch.randn(batch, hid, device=DEV, generator=gen))
    w0 = torch.randn(n_act + 1, hid, device=DEV, generator=gen) * 0.05
    b0 = torch.randn(n_act + 1, device=DEV, generator=gen) * 0.1
    act = torch.randint(0, n_act, (batch,), device=DEV, generator=gen).to(torch.uint8)
    adv = torch.randn(batch, device=DEV, generator=gen)
    ret = torch.randn(batch, device=DEV, generator=gen)
    p_k = torch.empty(batch, n_act, device=DEV)
    v_k = torch.empty(batch, device=DEV)
    L.pg_head_infer(h0, w0, b0, p_k, v_k)                     # the kernel's own probabilities
    rows = torch.arange(batch, device=DEV)
    a = act.long()
    old = torch.softmax(torch.log(p_k) + 0.25 * torch.randn(batch, n_act, device=DEV, generator=gen), 1)
    old[:128] = p_k[:128]                                     # ratio == 1 exactly
    adv[64:96] = 0.                                           # ... some of them with a zero advantage
    adv[200:232] = 0.                                         # and some ordinary rows too
    tiny = np.float32(1e-8)
    ratio_k = ((p_k[rows, a] + tiny) / (old[rows, a] + tiny))
    assert torch.all(ratio_k[:128] == 1.)
    lr_mult = torch.ones(1, device=DEV)
    c_v, c_e = 1.0, 0.01
    cases = [0.2, 0.0]
    below = ratio_k[(ratio_k > 0.6) & (ratio_k < 1.)]
    above = ratio_k[(ratio_k > 1.) & (ratio_k < 1.4)]
    cases += [float(np.float32(1.) - np.float32(below[0].item())), float(np.float32(above[0].item()) - np.float32(1.))]
    if tie == "math":
        cases = [0.2]       # ON a bound the mathematical derivative does not exist (torch splits the tie of its
        #                     max / min there; the kernel's "math" rule counts the bound as inside): not compared
    for clip in cases:
        lo, hi = np.float32(1.) - np.float32(clip), np.float32(1.) + np.float32(clip)
        n_on = int(((ratio_k == float(lo)) | (ratio_k == float(hi))).sum())
        assert n_on >= 1 or clip == 0.2, (clip, n_on)          # the constructed clips really put a row ON a bound
        h, w, bh = h0.clone().requires_grad_(), w0.clone().requires_grad_(), b0.clone().requires_grad_()
        out = h @ w.t() + bh
        prob, value = torch.softmax(out[:, :n_act], 1), out[:, n_act]
        ratio_t = (prob[rows, a] + 1e-8) / (old[rows, a] + 1e-8)
        ratio = ratio_t + (ratio_k - ratio_t).detach()         # the kernel's value, torch's derivative
        wgt = torch.full((batch,), 1. / batch, device=DEV)
        pi = -torch.sum(wgt * autograd_ref.ppo_surrogate(ratio, adv, float(clip), tie))
        vl = c_v * torch.sum(wgt * (value - ret) ** 2)
        el = -c_e * torch.sum(wgt * -torch.sum(prob * torch.log(prob + 1e-8), dim=1))
        gh, gw, gb = torch.autograd.grad(pi + vl + el, [h, w, bh])
        dout = torch.empty(batch, n_act + 1, device=DEV)
        dh = torch.empty(batch, hid, device=DEV)
        dw, db, loss4 = torch.empty_like(w0), torch.empty_like(b0), torch.zeros(4, device=DEV)
        L.pg_head_loss(h0, w0, b0, act, adv, ret, old, None, None, lr_mult, None, n_act, 1, clip, c_v, c_e,
                       dout, dh, dw, db, loss4, L.pg_head_workspace(DEV),
                       tie_rule=dict(theano=L.PPO_TIE_THEANO, math=L.PPO_TIE_MATH, both=L.PPO_TIE_BOTH)[tie])
        assert torch.allclose(loss4[:3], torch.stack([pi, vl, el]).detach(), rtol=1e-4, atol=1e-6), clip
        for got, want, name in ((dh, gh, "dh"), (dw, gw, "dw"), (db, gb, "db")):
            scale = max(want.abs().max().item(), 1e-6)
            assert torch.allclose(got, want, rtol=1e-3, atol=1e-5 * scale), (clip, name, (got - want).abs().max().item())


def test_staging_copy_and_ring_append(L):
    """arl_copy_bytes (the staging copy that is a kernel node: device <-> PINNED host memory, any size / alignment) and
    arl_ring_append (a captured graph's diagnostics into the slot the host can name by counting replays)."""
    rs = np.random.RandomState(3)
    for n in (1, 15, 16, 17, 4096, 10240 + 3, 1 << 20):
        src_h = torch.from_numpy(rs.randint(0, 256, size=n + 1, dtype=np.uint8)).pin_memory()
        dev = torch.zeros(n + 1, dtype=torch.uint8, device=DEV)
        back = torch.zeros(n + 1, dtype=torch.uint8).pin_memory()
        for off in (0, 1):                                   # 16-byte aligned and not
            dev.zero_()
            back.zero_()
            L.copy_bytes(dev[off:off + n], src_h[off:off + n])            # pinned host -> device
            L.copy_bytes(back[off:off + n], dev[off:off + n])             # device -> pinned host
            torch.cuda.synchronize()
            assert torch.equal(dev[off:off + n].cpu(), src_h[off:off + n]) and torch.equal(back[off:off + n], src_h[off:off + n])
            assert int(dev.sum().item()) == int(src_h[off:off + n].sum().item())   # nothing written next to the range
    with pytest.raises(RuntimeError, match="PINNED"):
        L.copy_bytes(dev, torch.zeros(n + 1, dtype=torch.uint8))
    # the same copy as a node of a hipGraph: every replay reads the pinned buffer's CURRENT contents
    host = torch.zeros(64, dtype=torch.float32).pin_memory()
    d = torch.zeros(64, device=DEV)
    L.copy_bytes(d, host)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        L.copy_bytes(d, host)
    for k in range(3):
        host.fill_(float(k + 1))
        g.replay()
        torch.cuda.synchronize()
        assert torch.all(d == k + 1)
    # ring: slot (counter % slots), counter advances on the device and stays reduced (never overflows)
    ring = torch.zeros((3, 5), device=DEV)
    count = torch.zeros(1, dtype=torch.int32, device=DEV)
    src = torch.zeros(5, device=DEV)
    g2 = torch.cuda.CUDAGraph()
    L.ring_append(src, ring, count)
    count.zero_()
    ring.zero_()
    torch.cuda.synchronize()
    with torch.cuda.graph(g2):
        L.ring_append(src, ring, count)
    for k in range(7):
        src.fill_(float(10 + k))
        g2.replay()
        torch.cuda.synchronize()
        assert int(count.item()) == (k + 1) % 3 and torch.all(ring[k % 3] == 10 + k)
        if k >= 1:
            assert torch.all(ring[(k - 1) % 3] == 10 + k - 1)         # the previous slot is untouched
    count.fill_(2 ** 31 - 1)                                          # a counter handed in out of range still names a slot
    g2.replay()
    torch.cuda.synchronize()
    assert int(count.item()) == ((2 ** 31 - 1) % 3 + 1) % 3 and torch.all(ring[(2 ** 31 - 1) % 3] == 16)
    with pytest.raises(ValueError):                                   # a slot must hold exactly src
        L.ring_append(src[:4], ring, count)
