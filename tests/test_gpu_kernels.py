"""Parity of the HIP kernels (through the C ABI, via hiast_amd.kernels) against the CPU oracle.
Integer/byte outputs: bit-exact.  Floating point: tolerance stated per test."""
import json

import numpy as np
import pytest
import torch

from hiast_amd import switches as SW

import synth
from oracle import cref, ias_ref, losses_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from hiast_amd import kernels
    return kernels


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


SHAPES = [(2, 19, 8, 16, 64, 128), (1, 19, 9, 17, 65, 129), (1, 9, 6, 6, 48, 48), (2, 19, 64, 128, 512, 1024),
          (1, 19, 5, 7, 5, 7), (1, 2, 3, 4, 100, 301), (1, 16, 33, 65, 257, 513)]


@pytest.mark.parametrize("shape", SHAPES)
def test_upsample_fwd_bit_exact(K, shape):
    B, C, h, w, H, W = shape
    x = synth.normal_f32(11, (B, C, h, w), 2.0)
    got = K.upsample_bilinear_ac_fwd(dev(x), H, W).cpu().numpy()
    want = cref.upsample_bilinear_ac(x, H, W)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("shape", SHAPES[:3] + SHAPES[4:6])
def test_upsample_bwd(K, shape):
    B, C, h, w, H, W = shape
    g = synth.normal_f32(12, (B, C, H, W))
    got = K.upsample_bilinear_ac_bwd(dev(g), h, w).cpu().numpy()
    want = cref.upsample_bilinear_ac_bwd(g, h, w)          # double accumulation
    assert np.allclose(got, want, rtol=1e-5, atol=1e-5)


def test_upsample_golden(K, golden):
    g = golden("upsample")
    for tag in "abc":
        B, C, h, w, H, W = [int(v) for v in g["shape_" + tag]]
        x = synth.normal_f32(100 + ord(tag), (B, C, h, w), 2.0)
        y = K.upsample_bilinear_ac_fwd(dev(x), H, W).cpu().numpy()
        assert np.abs(y - g["y_" + tag]).max() <= 2e-6 * np.abs(x).max()
        go = synth.normal_f32(200 + ord(tag), (B, C, H, W))
        gin = K.upsample_bilinear_ac_bwd(dev(go), h, w).cpu().numpy()
        assert np.allclose(gin, g["gin_" + tag], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("sigma", [3.0, 8.0])
def test_plabel_pass1_bit_exact(K, shape, sigma):
    B, C, h, w, H, W = shape
    z = synth.logits_lr(21, B, C, h, w, sigma)
    if C > 5:
        z[:, 1, : h // 2] = z[:, 0, : h // 2]     # ties: lower index must win
    mp, am, hist = K.plabel_pass1(dev(z), H, W)
    mp_o, am_o = cref.plabel_stage_a(z, H, W)
    assert np.array_equal(am.cpu().numpy(), am_o)
    assert np.array_equal(mp.cpu().numpy().view(np.uint32), mp_o.view(np.uint32))
    hist_o = cref.plabel_hist(mp_o, am_o, C)
    assert np.array_equal(hist.cpu().numpy().view(np.uint32), hist_o)
    assert int(hist_o.sum()) == B * H * W


def test_plabel_pass1_golden(K, golden):
    g = golden("stage_a")
    for tag in "abc":
        B, C, h, w, H, W = [int(v) for v in g["shape_" + tag]]
        z = synth.logits_lr(300 + ord(tag), B, C, h, w, float(g["sigma_" + tag]))
        if tag == "c":
            z[:, 1] = z[:, 0]
            z[:, 5] = z[:, 0]
        mp, am, _ = K.plabel_pass1(dev(z), H, W)
        assert np.array_equal(am.cpu().numpy(), g["argmax_" + tag])     # bit-exact argmax map
        assert np.allclose(mp.cpu().numpy(), g["maxprob_" + tag], rtol=4e-6, atol=0)


@pytest.mark.parametrize("hw", [(64, 128), (65, 129), (512, 1024), (7, 3)])
@pytest.mark.parametrize("with_thr", [True, False])
def test_plabel_pass2_bit_exact(K, hw, with_thr):
    H, W = hw
    C, B = 19, 3
    p, l = synth.probs_and_labels(31, B, H, W, C)
    thr = np.linspace(0.4, 0.97, C) if with_thr else None
    thr_up = None
    if with_thr:
        t32 = thr.astype(np.float32)
        t32 = np.where(t32.astype(np.float64) < thr, np.nextafter(t32, np.float32(np.inf)), t32)
        thr_up = dev(t32.astype(np.float32))
    plbl, count, sfx = K.plabel_pass2(dev(p), dev(l.astype(np.uint8)), thr_up, C)
    plbl_o, count_o, sfx_o = cref.plabel_select(p, l.astype(np.uint8), thr, C)
    assert np.array_equal(plbl.cpu().numpy(), plbl_o)
    assert np.array_equal(count.cpu().numpy(), count_o)
    assert np.array_equal(sfx.cpu().numpy().view(np.uint64), sfx_o)


def _loss_inputs(seed, B, C, h, w, H, W, p_ignore, dtype):
    z = synth.logits_lr(seed, B, C, h, w, 2.5)
    zt = synth.logits_lr(seed + 1, B, C, h, w, 2.5)
    plbl = synth.pseudo_labels(seed + 2, B, H, W, C, p_ignore, dtype)
    return z, zt, plbl


@pytest.mark.parametrize("region", ["ignored", "confident", "all"])
@pytest.mark.parametrize("shape", [(2, 19, 5, 9, 33, 65), (2, 19, 16, 32, 128, 256), (1, 9, 7, 7, 50, 50)])
@pytest.mark.parametrize("ldt", [np.uint8, np.int64])
def test_st_loss_fwd_bwd(K, region, shape, ldt):
    """4 sums + 3 counts and the gradient vs the torch-CPU oracle in float64; counts exact,
    sums 2e-5 rel, gradient 1e-4 rel of its max."""
    B, C, h, w, H, W = shape
    z, zt, plbl = _loss_inputs(700, B, C, h, w, H, W, 0.4, ldt)
    sums = K.st_loss_fwd(dev(z), dev(zt), dev(plbl), H, W, region)
    zl = torch.from_numpy(z).requires_grad_(True)
    pl = torch.from_numpy(plbl.astype(np.int64))
    s = losses_ref.st_loss_sums(zl, torch.from_numpy(zt), pl, (H, W), region)
    got = sums.cpu().numpy()
    want = np.array([s[k].item() for k in ("ce", "kld", "ent", "cst", "n_conf", "n_ign", "cst_cnt")])
    assert np.array_equal(got[4:7], want[4:7])
    assert np.allclose(got[:4], want[:4], rtol=2e-5)
    # gradient of  1.0*CE + 0.1*KLD + 1.0*ENT + 0.5*CST  with upstream grads (2, 3, 0.5, 1)
    L = losses_ref.st_losses(zl, torch.from_numpy(zt), pl, (H, W), region)
    up = [2.0, 3.0, 0.5, 1.0]
    names = ['target_seg_loss', 'kld_confident_loss', 'ent_ignored_loss', 'cst_loss']
    sum(u * L[n] for u, n in zip(up, names)).backward()
    coef = torch.tensor([up[0] * 1.0, up[1] * 0.1, up[2] * 1.0, up[3] * 0.5], dtype=torch.float32).cuda()
    d = K.st_loss_bwd(dev(z), dev(zt), dev(plbl), H, W, region, sums, coef).cpu().numpy()
    gref = zl.grad.numpy()
    assert np.abs(d - gref).max() <= 1e-4 * np.abs(gref).max()


def test_st_loss_golden_reference(K, golden):
    """against the reference's own compute_loss outputs (tests/golden/losses.npz)"""
    g = golden("losses")
    B, C, h, w, H, W = [int(v) for v in g["shape"]]
    for tag in ["mix", "conf", "all", "allign", "noign", "zeroq"]:
        cs = json.loads(str(g["cfg_" + tag]))
        z, zt, plbl = _loss_inputs(cs["seed"], B, C, h, w, H, W, cs["p_ignore"], np.int64)
        if tag == "zeroq":
            zt[:, 3] = -150.0
        sums = K.st_loss_fwd(dev(z), dev(zt), dev(plbl), H, W, cs["region"]).cpu().numpy()
        with np.errstate(invalid="ignore", divide="ignore"):
            vals = np.array([1.0 * sums[0] / sums[4], 0.1 * sums[1] / (C * sums[4]),
                             1.0 * sums[2] / (C * sums[5]), 0.5 * sums[3] / sums[6]])
        want = g["vals_" + tag]
        assert np.array_equal(np.isnan(vals), np.isnan(want)), tag
        ok = ~np.isnan(want)
        assert np.allclose(vals[ok], want[ok], rtol=2e-5), tag
        coef = torch.tensor([1.0 if ok[0] else 0, 0.1 if ok[1] else 0, 1.0 if ok[2] else 0, 0.5 if ok[3] else 0],
                            dtype=torch.float32).cuda()
        d = K.st_loss_bwd(dev(z), dev(zt), dev(plbl), H, W, cs["region"], dev(sums), coef).cpu().numpy()
        gr = g["grad_" + tag]
        assert np.abs(d - gr).max() <= 2e-4 * max(np.abs(gr).max(), 1e-12), tag


def test_st_loss_no_teacher(K):
    B, C, h, w, H, W = 2, 19, 8, 16, 64, 128
    z, _, plbl = _loss_inputs(900, B, C, h, w, H, W, 0.4, np.uint8)
    sums = K.st_loss_fwd(dev(z), None, dev(plbl), H, W, "ignored").cpu().numpy()
    s = losses_ref.st_loss_sums(torch.from_numpy(z), None, torch.from_numpy(plbl.astype(np.int64)), (H, W))
    assert np.allclose(sums[:3], [s["ce"].item(), s["kld"].item(), s["ent"].item()], rtol=2e-5)
    assert sums[3] == 0 and sums[6] == 0


def test_st_loss_deterministic(K):
    B, C, h, w, H, W = 2, 19, 16, 32, 128, 256
    z, zt, plbl = _loss_inputs(910, B, C, h, w, H, W, 0.4, np.uint8)
    a, b, c = dev(z), dev(zt), dev(plbl)
    coef = torch.tensor([1, .1, 1, .5], dtype=torch.float32).cuda()
    s1 = K.st_loss_fwd(a, b, c, H, W, "ignored")
    d1 = K.st_loss_bwd(a, b, c, H, W, "ignored", s1, coef)
    for _ in range(3):
        s2 = K.st_loss_fwd(a, b, c, H, W, "ignored")
        d2 = K.st_loss_bwd(a, b, c, H, W, "ignored", s2, coef)
        assert torch.equal(s1, s2) and torch.equal(d1, d2)


ASPP_SHAPES = [(1, 64, 9, 17, 19), (2, 128, 16, 32, 19), (1, 2048, 9, 17, 19), (3, 256, 30, 41, 9)]


def _aspp_inputs(seed, B, Cin, h, w, C):
    x = synth.normal_f32(seed, (B, Cin, h, w), 1.0)
    ws = [synth.normal_f32(seed + 1 + i, (C, Cin, 3, 3), 0.05) for i in range(4)]
    bs = [synth.normal_f32(seed + 11 + i, (C,), 0.1) for i in range(4)]
    return x, ws, bs


@pytest.mark.parametrize("shape", ASPP_SHAPES)
@pytest.mark.parametrize("dil", [(6, 12, 18, 24), (1, 2, 3, 5)])
def test_aspp_fwd(K, shape, dil):
    """fp32 MFMA vs the C oracle (double accumulation): |err| <= 1e-5 * Σ|w x| scale."""
    B, Cin, h, w, C = shape
    x, ws, bs = _aspp_inputs(40, B, Cin, h, w, C)
    wpack = K.aspp_pack_weights([dev(t) for t in ws], [dev(t) for t in bs])
    y = K.aspp_fwd(dev(x), wpack, C, dil).cpu().numpy()
    want = cref.aspp_fwd(x, ws, bs, dil)
    assert np.abs(y - want).max() <= 2e-5 * max(1.0, np.abs(want).max())


def test_aspp_fwd_golden(K, golden):
    g = golden("aspp")
    _, Cin, h, w, C = [int(v) for v in g["shape"]]
    x = synth.normal_f32(800, (1, Cin, h, w), 1.0)
    ws = [synth.normal_f32(810 + i, (C, Cin, 3, 3), 0.01) for i in range(4)]
    bs = [synth.normal_f32(820 + i, (C,), 0.1) for i in range(4)]
    wpack = K.aspp_pack_weights([dev(t) for t in ws], [dev(t) for t in bs])
    y = K.aspp_fwd(dev(x), wpack, C, (6, 12, 18, 24))
    assert np.allclose(y.cpu().numpy(), g["y"], rtol=1e-4, atol=1e-5)       # logits <= 1e-3 rel contract
    gy = synth.normal_f32(830, (1, C, h, w))
    dx = K.aspp_bwd_data(dev(gy), wpack, Cin, (6, 12, 18, 24)).cpu().numpy() if Cin % 256 == 0 else None
    assert np.allclose(dx[:, ::61], g["dx_sub"], rtol=1e-4, atol=1e-6)
    assert abs(dx.astype(np.float64).sum() - float(g["dx_sum"])) <= 1e-3 * np.abs(dx).sum() ** 0.5 + 1e-3
    dws, db = K.aspp_bwd_weight(dev(x), dev(gy), (6, 12, 18, 24))
    dws = np.stack([t.cpu().numpy() for t in dws])
    assert np.allclose(dws[:, :, ::97], g["dw_sub"], rtol=1e-4, atol=1e-5)
    assert np.allclose(db.cpu().numpy(), g["db"][0], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("shape", [(2, 256, 16, 32, 19), (1, 512, 9, 17, 19), (2, 256, 20, 70, 9)])
def test_aspp_bwd_vs_torch(K, shape):
    """dgrad / wgrad vs autograd of torch-CPU float64 convs (the oracle for a float kernel)."""
    B, Cin, h, w, C = shape
    dil = (6, 12, 18, 24)
    x, ws, bs = _aspp_inputs(50, B, Cin, h, w, C)
    gy = synth.normal_f32(60, (B, C, h, w))
    xt = torch.from_numpy(x).double().requires_grad_(True)
    wt = [torch.from_numpy(t).double().requires_grad_(True) for t in ws]
    bt = [torch.from_numpy(t).double().requires_grad_(True) for t in bs]
    y = sum(torch.nn.functional.conv2d(xt, wt[i], bt[i], 1, dil[i], dil[i]) for i in range(4))
    y.backward(torch.from_numpy(gy).double())
    wpack = K.aspp_pack_weights([dev(t) for t in ws], [dev(t) for t in bs])
    dx = K.aspp_bwd_data(dev(gy), wpack, Cin, dil).cpu().numpy()
    assert np.abs(dx - xt.grad.numpy()).max() <= 2e-5 * np.abs(xt.grad.numpy()).max()
    dws, db = K.aspp_bwd_weight(dev(x), dev(gy), dil)
    for i in range(4):
        ref = wt[i].grad.numpy()
        assert np.abs(dws[i].cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max(), i
    assert np.allclose(db.cpu().numpy(), bt[0].grad.numpy(), rtol=1e-5, atol=1e-5)


ASPP2_SHAPES = [(1, 128, 9, 17, 19), (2, 256, 16, 32, 19), (1, 2048, 9, 17, 19), (3, 256, 30, 41, 9), (2, 128, 5, 70, 32)]


def _bf16r(a):
    return torch.from_numpy(np.ascontiguousarray(a)).bfloat16().float().numpy()


def _aspp2_rounded_weights(ws):
    """what the bf16 GEMM multiplies with: every tap rounded to bf16, the four centre taps summed in fp32
    (ascending) first — expressed as four conv weights again so the direct-form oracle can consume them"""
    out = [_bf16r(w) for w in ws]
    centre = ((ws[0][:, :, 1, 1] + ws[1][:, :, 1, 1]) + ws[2][:, :, 1, 1]) + ws[3][:, :, 1, 1]
    out[0][:, :, 1, 1] = _bf16r(centre)
    for i in (1, 2, 3):
        out[i][:, :, 1, 1] = 0.0
    return out


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("shape", ASPP2_SHAPES)
@pytest.mark.parametrize("dil", [(6, 12, 18, 24), (1, 2, 3, 5)])
def test_aspp2_fwd_fp32_split(K, shape, dil):
    """channels-last GEMM + shift-add form, split-bf16 arithmetic, vs the C oracle (double accumulation) on the
    SAME fp32 inputs: fp32-class (contract on logits: 1e-3 relative)."""
    B, Cin, h, w, C = shape
    x, ws, bs = _aspp_inputs(140, B, Cin, h, w, C)
    wt, _, bias = K.aspp2_pack_weights([dev(t) for t in ws], [dev(t) for t in bs], need_dgrad=False)
    y = K.aspp2_fwd(_cl(dev(x)), wt, bias, dil).cpu().numpy()
    want = cref.aspp_fwd(x, ws, bs, dil)
    assert np.abs(y - want).max() <= 3e-5 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("shape", ASPP2_SHAPES)
def test_aspp2_fwd_split_planes(K, shape):
    """the same head fed with the trunk's split-plane output (LDS-DMA GEMM kernel)"""
    B, Cin, h, w, C = shape
    dil = (6, 12, 18, 24)
    x, ws, bs = _aspp_inputs(145, B, Cin, h, w, C)
    wt, _, bias = K.aspp2_pack_weights([dev(t) for t in ws], [dev(t) for t in bs], need_dgrad=False)
    xp = K.split_planes(dev(x).permute(0, 2, 3, 1).reshape(-1, Cin).contiguous()).view(B, h, w, 2 * Cin)
    y = K.aspp2_fwd(xp, wt, bias, dil, planes=2).cpu().numpy()
    want = cref.aspp_fwd(x, ws, bs, dil)
    assert np.abs(y - want).max() <= 3e-5 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("shape", ASPP2_SHAPES)
def test_aspp2_fwd_bf16(K, shape):
    """bf16 operands, fp32 accumulation: exact reference = the oracle on the bf16-rounded operands"""
    B, Cin, h, w, C = shape
    dil = (6, 12, 18, 24)
    x, ws, bs = _aspp_inputs(150, B, Cin, h, w, C)
    wt, _, bias = K.aspp2_pack_weights([dev(t) for t in ws], [dev(t) for t in bs], need_dgrad=False)
    y = K.aspp2_fwd(_cl(dev(x).bfloat16()), wt, bias, dil).cpu().numpy()
    want = cref.aspp_fwd(_bf16r(x), _aspp2_rounded_weights(ws), bs, dil)
    assert np.abs(y - want).max() <= 3e-5 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("shape", [(2, 256, 16, 32, 19), (1, 512, 9, 17, 19), (2, 256, 20, 70, 9), (1, 128, 64, 128, 19)])
def test_aspp2_bwd_vs_torch(K, shape):
    """mixed-precision dgrad / wgrad (bf16 operands, fp32 accumulate) vs autograd of torch-CPU float64 convs on
    the bf16-rounded operands; dx is delivered in bf16 (2^-9 relative rounding)."""
    B, Cin, h, w, C = shape
    dil = (6, 12, 18, 24)
    x, ws, bs = _aspp_inputs(160, B, Cin, h, w, C)
    gy = synth.normal_f32(170, (B, C, h, w))
    xt = torch.from_numpy(_bf16r(x)).double().requires_grad_(True)
    wt_ = [torch.from_numpy(t).double().requires_grad_(True) for t in _aspp2_rounded_weights(ws)]
    y = sum(torch.nn.functional.conv2d(xt, wt_[i], None, 1, dil[i], dil[i]) for i in range(4))
    y.backward(torch.from_numpy(_bf16r(gy)).double())
    _, wd, _ = K.aspp2_pack_weights([dev(t) for t in ws], [dev(t) for t in bs], need_dgrad=True)
    xd = _cl(dev(x).bfloat16())
    dx, dws, db = K.aspp2_bwd(xd, dev(gy), wd, dil)
    assert dx.shape == xd.shape and dx.permute(0, 2, 3, 1).is_contiguous()
    ref = xt.grad.numpy()
    err = np.abs(dx.float().cpu().numpy() - ref)
    assert (err <= 2.0 ** -8 * np.abs(ref) + 1e-4 * np.abs(ref).max()).all()
    for i in range(4):
        r = wt_[i].grad.numpy().copy()
        if i:
            r[:, :, 1, 1] = wt_[0].grad.numpy()[:, :, 1, 1]      # every branch's centre tap sees the same gradient
        assert np.abs(dws[i].cpu().numpy() - r).max() <= 5e-5 * np.abs(r).max(), i
    assert np.allclose(db.cpu().numpy(), gy.astype(np.float64).sum((0, 2, 3)), rtol=1e-5, atol=1e-5)
    # deterministic: fixed-order split reduction
    dx2, dws2, _ = K.aspp2_bwd(xd, dev(gy), wd, dil)
    assert torch.equal(dx, dx2) and all(torch.equal(a, b) for a, b in zip(dws, dws2))


def test_aspp_nhwc_autograd_matches_direct_form(K):
    """functional.aspp_nhwc (bf16, channels-last) against functional.aspp (exact fp32, NCHW) on the same data:
    forward within bf16 operand rounding, gradients within bf16 rounding of x / dY."""
    from hiast_amd import functional as HF
    B, Cin, h, w, C = 2, 256, 24, 40, 19
    x, ws, bs = _aspp_inputs(180, B, Cin, h, w, C)
    gy = dev(synth.normal_f32(190, (B, C, h, w)))
    outs = []
    for nhwc in (False, True):
        xt = dev(x).requires_grad_(True)
        wts = [dev(t).requires_grad_(True) for t in ws]
        bts = [dev(t).requires_grad_(True) for t in bs]
        if nhwc:
            y = HF.aspp_nhwc(_cl(xt.bfloat16()), wts, bts)
        else:
            y = HF.aspp(xt, wts, bts)
        y.backward(gy)
        outs.append((y.detach(), xt.grad, [t.grad for t in wts], [t.grad for t in bts]))
    (y0, gx0, gw0, gb0), (y1, gx1, gw1, gb1) = outs
    assert (y0 - y1).abs().max() <= 2e-2 * y0.abs().max()
    assert (gx0 - gx1.float()).abs().max() <= 2e-2 * gx0.abs().max()
    for a, b in zip(gw0, gw1):
        assert (a - b).abs().max() <= 2e-2 * a.abs().max()
    for a, b in zip(gb0, gb1):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-5)


def _planes_ref(a):
    """numpy fp32 -> (hi, lo) as fp32 arrays, the split the kernels use"""
    hi = _bf16r(a)
    lo = _bf16r(a - hi)
    return hi, lo


def test_split_planes_roundtrip(K):
    x = synth.normal_f32(300, (1000, 64), 3.0)
    x[0, :8] = [0.0, -0.0, 1.0, -1.0, 1e-30, 65504.0, 3.0e38, -2.5e-20]
    p = K.split_planes(dev(x))
    hi, lo = _planes_ref(x)
    assert tuple(p.shape) == (1000, 128) and p.dtype == torch.bfloat16
    got = np.sort(p.float().cpu().numpy(), axis=1)
    assert np.array_equal(got, np.sort(np.concatenate([hi, lo], 1), axis=1))      # same values, layout opaque
    back = K.merge_planes(p).cpu().numpy()
    assert np.array_equal(back, hi + lo)
    assert (np.abs(back - x) <= 2.0 ** -16 * np.abs(x)).all()


IGEMM_CASES = [  # B, H, W, Cin, Cout, taps, stride, dil, res
    (2, 9, 17, 64, 64, 1, 1, 1, False),
    (1, 16, 32, 256, 128, 1, 1, 1, True),
    (1, 8, 16, 1024, 256, 1, 1, 1, False),
    (2, 12, 20, 64, 64, 9, 1, 2, False),
    (1, 17, 23, 128, 256, 9, 2, 1, False),
    (1, 10, 40, 96, 128, 9, 1, 4, False),
    (3, 16, 16, 32, 192, 9, 1, 1, False),
]


def _igemm_ref(xf, wf, bn, resf, relu, stride, dil, taps):
    """float64 torch-CPU reference on the operand values the kernel multiplies"""
    xt = torch.from_numpy(xf).double().permute(0, 3, 1, 2)
    wt = torch.from_numpy(wf).double()
    y = torch.nn.functional.conv2d(xt, wt, None, stride if taps == 9 else 1, dil if taps == 9 else 0, dil if taps == 9 else 1)
    if bn is not None:
        g, b, mu, var, eps = bn
        sc = g / np.sqrt(var + eps)
        y = y * torch.from_numpy(sc).view(1, -1, 1, 1) + torch.from_numpy(b - mu * sc).view(1, -1, 1, 1)
    y = y.permute(0, 2, 3, 1)
    if resf is not None:
        y = y + torch.from_numpy(resf).double()
    if relu:
        y = y.clamp_min(0)
    return y.numpy()


def _mk_bn(seed, C):
    bn = torch.nn.BatchNorm2d(C).cuda().eval()
    with torch.no_grad():
        bn.weight.copy_(dev(1.0 + 0.2 * synth.normal_f32(seed, (C,))))
        bn.bias.copy_(dev(0.1 * synth.normal_f32(seed + 1, (C,))))
        bn.running_mean.copy_(dev(0.1 * synth.normal_f32(seed + 2, (C,))))
        bn.running_var.copy_(dev(np.abs(synth.normal_f32(seed + 3, (C,))) + 0.5))
    ref = tuple(t.detach().double().cpu().numpy() for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)) + (bn.eps,)
    return bn, ref


@pytest.mark.parametrize("case", IGEMM_CASES)
def test_igemm_split_planes(K, case):
    """fused conv + BN(eval) (+res) + ReLU on split planes (LDS-DMA staged, hi*hi + lo*hi + hi*lo): fp32-class"""
    B, H, W, Cin, Cout, taps, stride, dil, has_res = case
    kk = 3 if taps == 9 else 1
    x = synth.normal_f32(310, (B, H, W, Cin))
    w = synth.normal_f32(311, (Cout, Cin, kk, kk), (2.0 / (Cin * taps)) ** 0.5)
    bn, bnref = _mk_bn(312, Cout)
    xp = K.split_planes(dev(x).view(-1, Cin)).view(B, H, W, 2 * Cin)
    wp = K.pack_conv_weight(dev(w), 2)
    Ho, Wo = (H, W) if taps == 1 else ((H - 1) // stride + 1, (W - 1) // stride + 1)
    res = resp = None
    if has_res:
        res = synth.normal_f32(313, (B, Ho, Wo, Cout))
        resp = K.split_planes(dev(res).view(-1, Cout)).view(B, Ho, Wo, 2 * Cout)
    for relu in (True, False):
        y = K.igemm_bn_act(xp, wp, 2, bn, resp, relu, stride, dil)
        assert tuple(y.shape) == (B, Ho, Wo, 2 * Cout)
        got = K.merge_planes(y.view(-1, 2 * Cout)).view(B, Ho, Wo, Cout).cpu().numpy()
        xh, xl = _planes_ref(x)
        wh, wl = _planes_ref(w)
        rr = None if res is None else sum(_planes_ref(res))
        want = _igemm_ref(xh + xl, wh + wl, bnref, rr, relu, stride, dil, taps)
        tol = 3e-5 * max(1.0, np.abs(want).max())
        assert np.abs(got - want).max() <= tol, (np.abs(got - want).max(), tol)
        # the stored planes are a split of the fp32 value they merge to (same value after a re-split)
        v = K.merge_planes(y.view(-1, 2 * Cout))
        assert torch.equal(K.merge_planes(K.split_planes(v)), v)
    # fp32 output, no BN (the ASPP tap GEMM flavour)
    if not has_res:
        yf = K.igemm_bn_act(xp, wp, 2, None, None, False, stride, dil, out_f32=True).cpu().numpy()
        want = _igemm_ref(sum(_planes_ref(x)), sum(_planes_ref(w)), None, None, False, stride, dil, taps)
        assert np.abs(yf - want).max() <= 3e-5 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("case", IGEMM_CASES[1:5])
def test_igemm_bf16(K, case):
    """the same kernel on plain bf16 operands (one plane): exact reference = fp64 on the bf16-rounded operands,
    output rounded to bf16"""
    B, H, W, Cin, Cout, taps, stride, dil, has_res = case
    kk = 3 if taps == 9 else 1
    x = _bf16r(synth.normal_f32(320, (B, H, W, Cin)))
    w = synth.normal_f32(321, (Cout, Cin, kk, kk), (2.0 / (Cin * taps)) ** 0.5)
    bn, bnref = _mk_bn(322, Cout)
    xp = dev(x).bfloat16()
    wp = K.pack_conv_weight(dev(w), 1)
    Ho, Wo = (H, W) if taps == 1 else ((H - 1) // stride + 1, (W - 1) // stride + 1)
    res = resp = None
    if has_res:
        res = _bf16r(synth.normal_f32(323, (B, Ho, Wo, Cout)))
        resp = dev(res).bfloat16()
    y = K.igemm_bn_act(xp, wp, 1, bn, resp, True, stride, dil).float().cpu().numpy()
    want = _igemm_ref(x, _bf16r(w), bnref, res, True, stride, dil, taps)
    assert (np.abs(y - want) <= 2.0 ** -8 * np.abs(want) + 3e-5 * np.abs(want).max()).all()


@pytest.mark.parametrize("case", [(2, 13, 21, 256, 1024, 1, 1), (1, 16, 16, 64, 128, 1, 1), (2, 12, 20, 128, 256, 9, 2),
                                  (1, 9, 11, 512, 64, 1, 1)])
@pytest.mark.parametrize("mask", [False, True])
def test_igemm_gated_residual(K, case, mask):
    """data-gradient flavour of the igemm kernel (plain bf16 GEMM, no BN, no ReLU) with the gated residual of the
    identity hand-off: y = conv(x) + res * (gate > 0), the gate given as values (GATE = 1) or as the [M][Cout/8] bit
    mask the BN forward writes (GATE = 2); 256-row tiles with tail rows, every column-tile width; and the argument
    checks that keep the compile-time variants honest (no gate with ReLU / statistics with a residual)"""
    B, H, W, Cin, Cout, taps, dil = case
    kk = 3 if taps == 9 else 1
    x = _bf16r(synth.normal_f32(330, (B, H, W, Cin)))
    w = synth.normal_f32(331, (Cout, Cin, kk, kk), (2.0 / (Cin * taps)) ** 0.5)
    res = _bf16r(synth.normal_f32(332, (B, H, W, Cout)))
    gate = _bf16r(synth.normal_f32(333, (B, H, W, Cout)))
    gate[0, 0, :3, :] = 0.0                                     # exact zeros are closed
    xp, resp = dev(x).bfloat16(), dev(res).bfloat16()
    wp = K.pack_conv_weight(dev(w), 1)
    if mask:
        bits = np.packbits((gate > 0).reshape(B * H * W, Cout // 8, 8), axis=-1, bitorder="little").reshape(B * H * W, Cout // 8)
        g = dev(bits)
        assert g.dtype == torch.uint8
    else:
        g = dev(gate).bfloat16()
    y = K.igemm_bn_act(xp, wp, 1, None, resp, False, 1, dil, res_gate=g).float().cpu().numpy()
    want = _igemm_ref(x, _bf16r(w), None, res * (gate > 0), False, 1, dil, taps)
    assert (np.abs(y - want) <= 2.0 ** -8 * np.abs(want) + 3e-5 * np.abs(want).max()).all()
    plain = K.igemm_bn_act(xp, wp, 1, None, resp, False, 1, dil).float().cpu().numpy()        # GATE = 0: everything added
    want0 = _igemm_ref(x, _bf16r(w), None, res, False, 1, dil, taps)
    assert (np.abs(plain - want0) <= 2.0 ** -8 * np.abs(want0) + 3e-5 * np.abs(want0).max()).all()
    from hiast_amd._lib import HiastLibraryError
    with pytest.raises(HiastLibraryError):
        K.igemm_bn_act(xp, wp, 1, None, resp, True, 1, dil, res_gate=g)          # gate + ReLU: not a variant
    with pytest.raises(HiastLibraryError):
        K.igemm_bn_act(xp, wp, 1, None, resp, False, 1, dil, want_stats=True)    # statistics + residual: not a variant


@pytest.mark.parametrize("M_hw", [(1, 64, 128), (2, 50, 77), (1, 64, 65)])
def test_xconv_expanding_1x1(K, M_hw, monkeypatch):
    """K9e (xconv.hip): the register-resident-weight kernel that takes the 256 -> 1024 1x1 launches of layer3 over from
    the tile kernel — every epilogue variant against float64 on the bf16 operands AND against the tile kernel
    (HIAST_XCONV=0) on the same inputs; ragged M (tail panel, rows beyond M), statistics of the stored values"""
    B, H, W = M_hw
    Cin, Cout = 256, 1024
    x = _bf16r(synth.normal_f32(340, (B, H, W, Cin)))
    w = synth.normal_f32(341, (Cout, Cin, 1, 1), (2.0 / Cin) ** 0.5)
    res = _bf16r(synth.normal_f32(342, (B, H, W, Cout)))
    gate = synth.normal_f32(343, (B, H, W, Cout))
    bits = dev(np.packbits((gate > 0).reshape(B * H * W, Cout // 8, 8), axis=-1, bitorder="little").reshape(B * H * W, Cout // 8))
    bn, bnref = _mk_bn(344, Cout)
    xp, resp = dev(x).bfloat16(), dev(res).bfloat16()
    wp = K.pack_conv_weight(dev(w), 1)
    assert K._lib.load().hiast_igemm_stats_rows(B * H * W, Cin, Cout, 1, 1) != (B * H * W + 255) // 256 or B * H * W < 4096
    cases = [("plain", dict(bn=None, res=None, relu=False), None),
             ("bn_relu", dict(bn=bn, res=None, relu=True), bnref),
             ("bn_res_relu", dict(bn=bn, res=resp, relu=True), bnref),
             ("bn_res", dict(bn=bn, res=resp, relu=False), bnref),
             ("res", dict(bn=None, res=resp, relu=False), None),
             ("gated", dict(bn=None, res=resp, relu=False, res_gate=bits), None)]
    for name, kw, bref in cases:
        args = (xp, wp, 1, kw["bn"], kw["res"], kw["relu"], 1, 1)
        extra = {k: v for k, v in kw.items() if k == "res_gate"}
        y = K.igemm_bn_act(*args, **extra)
        monkeypatch.setenv("HIAST_XCONV", "0")
        y_tile = K.igemm_bn_act(*args, **extra)
        monkeypatch.delenv("HIAST_XCONV")
        rr = None if kw["res"] is None else (res * (gate > 0) if name == "gated" else res)
        want = _igemm_ref(x, _bf16r(w), bref, rr, kw["relu"], 1, 1, 1)
        got = y.float().cpu().numpy()
        assert (np.abs(got - want) <= 2.0 ** -8 * np.abs(want) + 3e-5 * np.abs(want).max()).all(), name
        # the two kernels accumulate k in different orders: equal up to one bf16 rounding of a ~1e-6 difference
        d = (y.float() - y_tile.float()).abs()
        assert float((d > 2.0 ** -7 * y_tile.float().abs() + 1e-5).float().mean()) == 0.0, name
        assert float((d > 0).float().mean()) < 0.02, name
    y, part = K.igemm_bn_act(xp, wp, 1, None, None, False, 1, 1, want_stats=True)
    yf = y.float().view(-1, Cout).double()
    sums = part.double().sum(0)
    assert torch.allclose(sums[:, 0], yf.sum(0), rtol=1e-5, atol=1e-3) and torch.allclose(sums[:, 1], (yf * yf).sum(0), rtol=1e-5)
    s64 = K.bn_nhwc_stats_from_partial(part)
    assert torch.allclose(s64[:, 0], yf.sum(0), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("shape", [(2, 64, 9, 17), (3, 256, 16, 24), (1, 2048, 8, 12), (2, 1024, 5, 7)])
@pytest.mark.parametrize("res,relu", [(False, True), (True, True), (False, False)])
def test_bn_nhwc_matches_fp64(K, shape, res, relu):
    """training-mode BN (+res)(+ReLU) on channels-last bf16, forward and backward, vs float64 torch autograd on the
    same bf16-rounded inputs (outputs are bf16: 2^-9 relative rounding)"""
    from hiast_amd import functional as HF
    B, C, H, W = shape
    x = _bf16r(synth.normal_f32(400, shape, 2.0) + 0.3)
    r = _bf16r(synth.normal_f32(401, shape)) if res else None
    gy = _bf16r(synth.normal_f32(402, shape))
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(dev(1.0 + 0.2 * synth.normal_f32(403, (C,))))
        bn.bias.copy_(dev(0.1 * synth.normal_f32(404, (C,))))
    xt = _cl(dev(x).bfloat16()).requires_grad_(True)
    rt = _cl(dev(r).bfloat16()).requires_grad_(True) if res else None
    y = HF.bn_act(xt, bn, rt, relu)
    assert y.dtype == torch.bfloat16 and y.permute(0, 2, 3, 1).is_contiguous()
    y.backward(_cl(dev(gy).bfloat16()))
    # reference
    xd = torch.from_numpy(x).double().requires_grad_(True)
    rd = torch.from_numpy(r).double().requires_grad_(True) if res else None
    g, b = bn.weight.detach().double().cpu(), bn.bias.detach().double().cpu()
    mu = xd.mean((0, 2, 3), keepdim=True)
    var = xd.var((0, 2, 3), unbiased=False, keepdim=True)
    yd = (xd - mu) / torch.sqrt(var + bn.eps) * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
    if res:
        yd = yd + rd
    if relu:
        yd = torch.relu(yd)
    yd.backward(torch.from_numpy(gy).double())
    tol = lambda ref: 2.0 ** -7 * ref.abs() + 2e-3 * ref.abs().max()
    assert ((y.double().cpu() - yd.detach()).abs() <= tol(yd.detach())).all()
    assert ((xt.grad.double().cpu() - xd.grad).abs() <= tol(xd.grad)).all()
    if res:
        assert ((rt.grad.double().cpu() - rd.grad).abs() <= tol(rd.grad)).all()
    # running statistics follow torch's update rule
    n = B * H * W
    assert torch.allclose(bn.running_mean.double().cpu(), 0.1 * mu.detach().flatten(), atol=1e-5)
    assert torch.allclose(bn.running_var.double().cpu(), 0.9 + 0.1 * var.detach().flatten() * n / (n - 1), rtol=1e-5)


@pytest.mark.parametrize("cfg", [(2, 64, 64, 10, 18, 1, 1, 1), (1, 256, 128, 12, 20, 3, 1, 2), (2, 128, 128, 16, 16, 3, 2, 1),
                                 (1, 512, 256, 9, 11, 3, 1, 4), (1, 1024, 256, 8, 8, 1, 1, 1),
                                 (2, 256, 256, 16, 16, 3, 2, 1), (2, 256, 512, 20, 33, 3, 1, 2), (3, 256, 1024, 33, 21, 1, 1, 1)])
def test_conv_nhwc_autograd_vs_fp64(K, cfg, monkeypatch):
    """_ConvNhwcFn (igemm forward / data gradient; weight gradient by hiast_conv_wgrad_nhwc where the shape allows,
    else by the library) vs float64 autograd on bf16-rounded data"""
    from hiast_amd import functional as HF
    monkeypatch.delenv("HIAST_LIB_WGRAD3", raising=False)  # the 3x3 form of the own weight-gradient kernel is the default
    B, Cin, Cout, H, W, k, stride, dil = cfg
    conv = torch.nn.Conv2d(Cin, Cout, k, stride=stride, padding=dil if k == 3 else 0, dilation=dil if k == 3 else 1,
                           bias=False).cuda()
    with torch.no_grad():
        conv.weight.copy_(dev(synth.normal_f32(410, tuple(conv.weight.shape), (2.0 / (Cin * k * k)) ** 0.5)))
    x = _bf16r(synth.normal_f32(411, (B, Cin, H, W)))
    xt = _cl(dev(x).bfloat16()).requires_grad_(True)
    assert HF.conv_nhwc_ok(xt, conv)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = HF.conv_nhwc(xt, conv)
    gy = _bf16r(synth.normal_f32(412, tuple(y.shape)))
    y.backward(_cl(dev(gy).bfloat16()))
    xd = torch.from_numpy(x).double().requires_grad_(True)
    wd = torch.from_numpy(_bf16r(conv.weight.detach().cpu().numpy())).double().requires_grad_(True)
    yd = torch.nn.functional.conv2d(xd, wd, None, stride, dil if k == 3 else 0, dil if k == 3 else 1)
    yd.backward(torch.from_numpy(gy).double())
    tol = lambda ref: 2.0 ** -7 * ref.abs() + 2e-3 * ref.abs().max()
    assert ((y.double().cpu() - yd.detach()).abs() <= tol(yd.detach())).all()
    assert ((xt.grad.double().cpu() - xd.grad).abs() <= tol(xd.grad)).all()
    assert conv.weight.grad.dtype == torch.float32
    assert ((conv.weight.grad.double().cpu() - wd.grad).abs() <= tol(wd.grad)).all()
    # statistics epilogue: per-256-row-block sums of the STORED outputs, and the BN forward fed by them
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y2, partial = HF.conv_nhwc(xt.detach(), conv, want_stats=True)
    assert torch.equal(y2, y.detach())
    yf = y2.float().permute(0, 2, 3, 1).reshape(-1, Cout).double()
    nb = (yf.shape[0] + 255) // 256
    assert tuple(partial.shape) == (nb, Cout, 2)
    for b in range(nb):
        blk = yf[b * 256:(b + 1) * 256]
        assert torch.allclose(partial[b, :, 0].double(), blk.sum(0), rtol=1e-4, atol=1e-3)
        assert torch.allclose(partial[b, :, 1].double(), (blk * blk).sum(0), rtol=1e-4, atol=1e-3)
    s_ep = K.bn_nhwc_stats_from_partial(partial)
    s_pass = K.bn_nhwc_stats(y2)
    assert torch.allclose(s_ep, s_pass, rtol=1e-5, atol=1e-3)


def test_wgrad_side_stream_gives_identical_gradients(K):
    """weight gradients issued on the side stream (single-process trainers) == the main-stream ones after the join"""
    from hiast_amd import functional as HF
    from hiast_amd.sseg.models.modules.resnet import Bottleneck
    torch.manual_seed(1)
    blk = Bottleneck(512, 256, 1, 2).cuda().train()          # 256/512/1024-channel 1x1s: the own weight-gradient kernel
    x0 = _cl(dev(_bf16r(synth.normal_f32(960, (2, 1024, 16, 24)))).bfloat16())
    gy = _cl(dev(_bf16r(synth.normal_f32(961, (2, 1024, 16, 24)))).bfloat16())
    blk = Bottleneck(1024, 256, 1, 2).cuda().train()
    grads = []
    for overlap in (False, True):
        blk.zero_grad(set_to_none=True)
        HF.enable_wgrad_overlap(overlap)
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = blk(x0.clone().requires_grad_(True) * 1.0)
            y.backward(gy)
        finally:
            HF.enable_wgrad_overlap(False)
        HF.wgrad_stream_join()
        grads.append([blk.conv1.weight.grad.clone(), blk.conv3.weight.grad.clone()])
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])   # own kernel: deterministic


def test_h2d_async_uploads_through_pinned_staging(K):
    """kernels.h2d_async: values arrive, dtype/shape kept, an existing device table is refilled in place, and the
    host array may be overwritten right after the call (the bytes were staged)"""
    a = np.arange(19, dtype=np.float32) * 0.5
    t = K.h2d_async(a, torch.device("cuda"))
    a[:] = -1.0
    assert t.dtype == torch.float32 and tuple(t.shape) == (19,)
    assert torch.equal(t.cpu(), torch.arange(19, dtype=torch.float32) * 0.5)
    table = torch.zeros(64, dtype=torch.uint8, device="cuda")
    ptr = table.data_ptr()
    rec = np.arange(8, dtype=np.int64)
    out = K.h2d_async(rec.view(np.uint8).reshape(-1), table.device, out=table)
    rec[:] = 0
    assert out.data_ptr() == ptr
    assert np.array_equal(table.cpu().numpy().view(np.int64), np.arange(8, dtype=np.int64))


@pytest.mark.parametrize("cfg", [(2, 256, 256, 64, 128, 3, 2), (1, 512, 512, 64, 128, 3, 4), (2, 1024, 256, 64, 128, 1, 1),
                                 (1, 256, 1024, 61, 77, 1, 1), (1, 256, 256, 50, 70, 3, 1)])
def test_conv_wgrad_many_pixel_ranges(K, cfg):
    """hiast_conv_wgrad_nhwc at sizes where the pixel index is split over many blocks (28-64 ranges): the XCD-aware
    block order with a block count that is not a multiple of 8, ragged last ranges, the coalesced fixed-order reduce;
    against the fp32 weight gradient of the same bf16-rounded operands; bitwise repeatable"""
    B, Cin, Cout, H, W, k, dil = cfg
    x = dev(_bf16r(synth.normal_f32(970, (B, H, W, Cin)))).bfloat16()
    dy = dev(_bf16r(synth.normal_f32(971, (B, H, W, Cout)))).bfloat16()
    dw = K.conv_wgrad_nhwc(dy, x, k, 1, dil)
    assert dw.dtype == torch.float32 and tuple(dw.shape) == (Cout, Cin, k, k)
    pad = dil if k == 3 else 0
    # the library's fp32 weight gradient of the same (exactly representable) operands: both sides only differ by their
    # fp32 accumulation order; a lost tap / pixel range / block would be off by O(max|dW|)
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).float(), (Cout, Cin, k, k), dy.permute(0, 3, 1, 2).float(),
                                      stride=1, padding=pad, dilation=dil if k == 3 else 1)
    err = (dw - ref).abs().max().item()
    assert err <= 1e-4 * ref.abs().max().item(), (err, ref.abs().max().item())
    assert torch.equal(dw, K.conv_wgrad_nhwc(dy, x, k, 1, dil))


def test_bottleneck_identity_handoff_matches_autograd_add(K, monkeypatch):
    """identity block, channels-last training path: the gradient of the identity branch added in conv1's data-gradient
    epilogue (gated by the block output) vs the plain autograd formulation (masked copy + add kernel)"""
    from hiast_amd.sseg.models.modules.resnet import Bottleneck
    torch.manual_seed(0)
    blk = Bottleneck(256, 64, 1, 2).cuda().train()
    x0 = _cl(dev(_bf16r(synth.normal_f32(950, (2, 256, 24, 40)))).bfloat16())
    gy = _cl(dev(_bf16r(synth.normal_f32(951, (2, 256, 24, 40)))).bfloat16())
    outs = []
    for off in ("1", "0"):
        monkeypatch.setitem(SW.SWITCHES, "HIAST_NO_IDT_HANDOFF", off == "1")
        blk.zero_grad()
        # the block input is itself the output of an op (as in the trunk), so that its gradient is what autograd delivers
        src = x0.clone().requires_grad_(True)
        xin = src * 1.0
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = blk(xin)
        y.backward(gy)
        outs.append((y.detach().float(), src.grad.float(), blk.conv1.weight.grad.clone(), blk.conv3.weight.grad.clone()))
    (y0, g0, a0, c0), (y1, g1, a1, c1) = outs
    assert torch.equal(y0, y1)
    # one bf16 rounding of (data gradient + masked identity gradient) instead of two: differences are ulps of the
    # LARGER addend, so the bound is relative to the tensor's scale
    assert ((g0 - g1).abs() <= 2.0 ** -6 * g0.abs() + 2.0 ** -7 * g0.abs().max()).all()
    assert float((g0 - g1).abs().mean()) <= 2e-3 * float(g0.abs().mean())
    # weight gradients do not depend on the hand-off (64-channel shapes: library kernels, bf16 results, atomics)
    assert torch.allclose(a0, a1, rtol=1e-2, atol=1e-2 * float(a0.abs().max()))
    assert torch.allclose(c0, c1, rtol=1e-2, atol=1e-2 * float(c0.abs().max()))


def test_training_trunk_channels_last_as_accurate_as_nchw_path(K):
    """mixed-precision training forward/backward of the whole DeepLab.  A random-init train-mode ResNet-101 amplifies
    bf16 rounding noise layer by layer (both bf16 paths end ~0.8 of max away from the fp32 forward), so the
    channels-last path (own conv + BN kernels) is checked to be AS CLOSE to the fp32 path as the NCHW path
    (library convs + NCHW BN kernels) is, at every stage, and to agree with it while the noise is still small."""
    import os
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import SEG_MODEL
    from make_golden import seeded_state_dict
    x = torch.from_numpy(synth.normal_f32(905, (2, 3, 97, 129))).cuda()
    res = {}
    for mode in ("fp32", "nchw", "nhwc"):
        m = SEG_MODEL["DeepLab_V2"](19, 256)
        m.load_state_dict(seeded_state_dict(m, 9100))
        m = m.cuda().train()
        acts = {}

        def hook(name, acts=acts):
            return lambda mod, i, o: acts.__setitem__(name, o.detach().float().contiguous())
        for n in ("layer1", "layer2", "layer3", "layer4"):
            getattr(m.backbone, n).register_forward_hook(hook(n))
        os.environ["HIAST_TRAIN_NCHW"] = "0" if mode == "nhwc" else "1"
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode != "fp32"):
                pred, feat = m(x)
            if mode == "nhwc":
                assert feat.dtype == torch.bfloat16 and feat.permute(0, 2, 3, 1).is_contiguous()
            pred.float().square().mean().backward()
        finally:
            os.environ.pop("HIAST_TRAIN_NCHW", None)
        acts["pred"] = pred.detach().float()
        acts["g_aspp"] = m.aspp.conv2d_list[1].weight.grad.clone()
        acts["g_l4"] = m.backbone.layer4[2].conv3.weight.grad.clone()
        acts["rv"] = m.backbone.layer2[3].bn3.running_var.clone()
        res[mode] = acts

    def d(u, v):
        return float((u - v).abs().max() / u.abs().max())
    for k in ("layer1", "layer2", "layer3", "layer4", "pred"):
        e_nchw, e_nhwc = d(res["fp32"][k], res["nchw"][k]), d(res["fp32"][k], res["nhwc"][k])
        if e_nchw < 0.2:        # while the rounding noise is still small: as close to fp32 as the library path
            assert e_nhwc <= 1.3 * e_nchw + 2e-3, (k, e_nchw, e_nhwc)
        else:                   # deep in the amplified regime (both paths 0.5 - 0.8 of the maximum away from fp32, the library
            # path's figure moves with the algorithm MIOpen picks on the box: 0.52 ... 0.8): only "as chaotic as", the
            # arithmetic itself is pinned by the fp64 kernel tests and the gated-oracle step test
            assert e_nhwc <= 2.0 * e_nchw + 0.2, (k, e_nchw, e_nhwc)
    # (round 4: the two paths no longer share the stem convolution — K9k against the library's — so their bf16 roundings are
    # independent from the first layer on, and each is ~e away from the fp32 forward: what they may differ by from EACH OTHER is
    # the sum of those two distances; measured 0.025 / 0.10 of the maximum at layer1 / layer2)
    for k in ("layer1", "layer2"):
        e_nchw, e_nhwc = d(res["fp32"][k], res["nchw"][k]), d(res["fp32"][k], res["nhwc"][k])
        assert d(res["nchw"][k], res["nhwc"][k]) <= 1.25 * (e_nchw + e_nhwc) + 2e-3, (k, e_nchw, e_nhwc)
    assert d(res["nchw"]["layer1"], res["nhwc"]["layer1"]) < 4e-2
    assert torch.allclose(res["nchw"]["rv"], res["nhwc"]["rv"], rtol=5e-2, atol=1e-4)
    for k in ("g_aspp", "g_l4"):       # gradients next to the loss: as aligned with fp32 as the NCHW path's
        cos = lambda a, b: float(torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0))
        assert torch.isfinite(res["nhwc"][k]).all()
        assert cos(res["fp32"][k], res["nhwc"][k]) >= cos(res["fp32"][k], res["nchw"][k]) - 0.1, k


def test_fused_adam_ema_cosine_vs_reference_golden(K, golden):
    """the trajectory the reference's Adam(wd) + update_ema_model + cosine schedule produce (fixture G10), with the
    one-launch HIP Adam step, the HIP EMA kernel and this package's scheduler"""
    from types import SimpleNamespace as ns
    from hiast_amd.utils import utils
    from hiast_amd.sseg.models.modules.schedulers import build_scheduler
    g = golden("ema_optim")
    shapes = [(7, 5), (11,), (3, 2, 3, 3)]
    p = [torch.nn.Parameter(dev(synth.normal_f32(1200 + i, s))) for i, s in enumerate(shapes)]
    e = [q.detach().clone() for q in p]
    opt = utils.FusedAdam([{"params": p[:2], "lr": 3e-6}, {"params": p[2:], "lr": 3e-5}], betas=(0.9, 0.999),
                          weight_decay=0.0005)
    cfg = ns(train=ns(total_iter=10, lr=3e-6, lr_scheduler=ns(type="Cosine")))
    sc = build_scheduler(cfg, opt)
    plan = K.EmaPlan(e, [q.data for q in p])
    for step in range(3):
        opt.zero_grad()
        for i, q in enumerate(p):
            q.grad = dev(synth.normal_f32(1300 + 10 * step + i, shapes[i]))
        opt.step()
        K.ema_update(plan, 0.999)
        sc.step()
        got_p = np.concatenate([q.detach().cpu().numpy().ravel() for q in p])
        got_e = np.concatenate([k.cpu().numpy().ravel() for k in e])
        assert np.allclose(got_p, g["p"][step], rtol=1e-6, atol=1e-9)
        assert np.allclose(got_e, g["e"][step], rtol=1e-6, atol=1e-9)
        assert np.allclose([opt.param_groups[0]["lr"], opt.param_groups[1]["lr"]], g["lr"][step], rtol=1e-9)


def test_fused_adam_matches_torch_adam(K):
    from hiast_amd.utils import utils
    shapes = [(70001,), (64, 256, 3, 3), (19,), (5, 7)]
    pa = [torch.nn.Parameter(dev(synth.normal_f32(1400 + i, s))) for i, s in enumerate(shapes)]
    pb = [torch.nn.Parameter(q.detach().clone()) for q in pa]
    oa = utils.FusedAdam([{"params": pa[:2], "lr": 3e-4}, {"params": pa[2:], "lr": 3e-3}], weight_decay=0.0005)
    ob = torch.optim.Adam([{"params": pb[:2], "lr": 3e-4}, {"params": pb[2:], "lr": 3e-3}], weight_decay=0.0005)
    for step in range(4):
        for i, (a, b) in enumerate(zip(pa, pb)):
            gr = dev(synth.normal_f32(1500 + 10 * step + i, shapes[i]))
            a.grad = None if (step == 1 and i == 3) else gr.clone()       # a parameter that skips a step
            b.grad = None if (step == 1 and i == 3) else gr.clone()
        oa.step()
        ob.step()
        for a, b in zip(pa, pb):
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-8)
    sa, sb = oa.state_dict()["state"], ob.state_dict()["state"]
    for k in sb:
        assert float(sa[k]["step"]) == float(sb[k]["step"])
        # (torch's foreach path forms m by lerp / mul+add in another association: a few ulps of the LARGER term)
        assert torch.allclose(sa[k]["exp_avg"], sb[k]["exp_avg"], rtol=1e-5, atol=1e-6 * float(sb[k]["exp_avg"].abs().max()))
        assert torch.allclose(sa[k]["exp_avg_sq"], sb[k]["exp_avg_sq"], rtol=1e-5,
                              atol=1e-6 * float(sb[k]["exp_avg_sq"].abs().max()))


def test_normalize_u8_bit_exact_vs_host_transform(K):
    """device ToTensor + Normalize == the DataLoader-worker transform (torchvision semantics), bit for bit"""
    from hiast_amd.sseg.datasets import utils as du
    r = synth.rng(77)
    img = r.integers(0, 256, size=(3, 37, 53, 3), dtype=np.uint8)
    img[0, 0, :8, 0] = [0, 1, 2, 127, 128, 254, 255, 3]
    got = K.normalize_u8(dev(img), du.MEAN, du.STD).cpu()
    want = torch.stack([du._img_to_tensor(img[b], du.MEAN, du.STD) for b in range(3)])
    assert got.dtype == torch.float32 and tuple(got.shape) == (3, 3, 37, 53)
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    t_img, _ = du.transform(img[1], np.zeros((37, 53), np.uint8), raw_u8=True)
    assert t_img.dtype == torch.uint8 and tuple(t_img.shape) == (37, 53, 3)


def test_pack_plan_equals_single_packs(K):
    """the one-launch multi-tensor weight pack == the per-weight packs, forward and adjoint forms, and it refreshes only
    when a weight's version counter has moved"""
    ws = [torch.nn.Parameter(dev(synth.normal_f32(1600 + i, shp))) for i, shp in
          enumerate([(64, 64, 1, 1), (128, 64, 3, 3), (256, 128, 1, 1), (64, 256, 3, 3)])]
    for PL in (1, 2):
        plan = K.PackPlan(ws, PL, [True, False, True, True])
        assert plan.refresh() and not plan.refresh()
        for i, w in enumerate(ws):
            assert torch.equal(plan.wp[i], K.pack_conv_weight(w, PL))
            if plan.wpt[i] is not None:
                assert torch.equal(plan.wpt[i], K.pack_conv_weight(w, PL, transpose=True))
            else:
                assert i == 1
        with torch.no_grad():
            ws[2].mul_(2.0)
        assert plan.refresh()
        assert torch.equal(plan.wp[2], K.pack_conv_weight(ws[2], PL)) and torch.equal(plan.wp[0], K.pack_conv_weight(ws[0], PL))
        with torch.no_grad():
            ws[2].mul_(0.5)


def _pack_ref(w, PL, transpose):
    """the packed operand format spelled out with torch: rows x taps x (PL * reduction channels); bf16 hi (and
    lo = bf16(v - hi)) in slabs of 32 channels [hi32 | lo32] for split planes; the adjoint form swaps the channel
    roles and flips the taps"""
    N, Kc, kh, kw = w.shape
    if transpose:
        w = w.flip(2, 3).permute(1, 0, 2, 3)
        N, Kc = Kc, N
    v = w.permute(0, 2, 3, 1).reshape(N, kh * kw, Kc).float()
    hi = v.bfloat16()
    if PL == 1:
        return hi
    lo = (v - hi.float()).bfloat16()
    return torch.stack([hi.view(N, kh * kw, Kc // 32, 32), lo.view(N, kh * kw, Kc // 32, 32)], 3).reshape(N, kh * kw, 2 * Kc)


@pytest.mark.parametrize("shape", [(64, 64, 1, 1), (256, 128, 3, 3), (128, 320, 1, 1), (192, 64, 3, 3), (640, 2048, 1, 1),
                                   (48, 96, 3, 3)])
def test_pack_conv_weight_layout(K, shape):
    """tiled (multiples of 64) and elementwise (anything else) packing kernels against the layout definition"""
    w = dev(synth.normal_f32(1700 + shape[0], shape))
    for PL in (1, 2):
        if PL == 2 and (shape[0] % 32 or shape[1] % 32):
            continue
        assert torch.equal(K.pack_conv_weight(w, PL), _pack_ref(w, PL, False))
        assert torch.equal(K.pack_conv_weight(w, PL, transpose=True), _pack_ref(w, PL, True))
        f, a = K.pack_conv_weight(w, PL, both=True)
        assert torch.equal(f, _pack_ref(w, PL, False)) and torch.equal(a, _pack_ref(w, PL, True))


def test_ema_bit_exact(K):
    shapes = [(7, 5), (70001,), (3, 2, 3, 3), (64, 2048, 1, 1)]
    e = [synth.normal_f32(70 + i, s) for i, s in enumerate(shapes)]
    p = [synth.normal_f32(80 + i, s) for i, s in enumerate(shapes)]
    et = [dev(a) for a in e]
    pt = [dev(a) for a in p]
    plan = K.EmaPlan(et, pt)
    for _ in range(2):
        K.ema_update(plan, 0.999)
        for a, b in zip(e, p):
            cref.ema_update(a.reshape(-1), b.reshape(-1), 0.999)
    for a, t in zip(e, et):
        assert np.array_equal(a.view(np.uint32), t.cpu().numpy().view(np.uint32))


def test_confusion_hist_exact(K):
    g = synth.rng(90)
    pred = g.integers(0, 19, size=(2, 300, 400), dtype=np.int64)
    tgt = g.integers(0, 19, size=(2, 300, 400), dtype=np.int64)
    tgt[g.random((2, 300, 400)) < 0.2] = 255
    i, ap, at = K.confusion_hist(dev(pred), dev(tgt), 19)
    io, apo, ato = cref.confusion_hist(pred, tgt, 19)
    assert np.array_equal(i.cpu().numpy(), io) and np.array_equal(ap.cpu().numpy(), apo)
    assert np.array_equal(at.cpu().numpy(), ato)


def test_no_cpu_fallback(K):
    from hiast_amd._lib import HiastLibraryError
    with pytest.raises(HiastLibraryError):
        K.upsample_bilinear_ac_fwd(torch.zeros(1, 1, 2, 2), 4, 4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 64, 16, 32), (3, 10, 7, 9), (8, 256, 64, 128)])
@pytest.mark.parametrize("res,relu", [(False, True), (True, True), (False, False)])
def test_bn_act_train_fwd_bwd_vs_torch(K, dtype, shape, res, relu):
    """fused BN(+res)(+ReLU) with batch statistics, forward, running-stat update and backward, vs
    torch.nn.BatchNorm2d in float64 on the CPU."""
    from hiast_amd import functional as HF
    B, C, Hh, Ww = shape
    x = synth.normal_f32(101, shape, 2.0) + 0.5
    r = synth.normal_f32(102, shape, 1.0) if res else None
    gy = synth.normal_f32(103, shape, 1.0)
    gam = 1 + synth.normal_f32(104, (C,), 0.2)
    bet = synth.normal_f32(105, (C,), 0.2)
    bn = torch.nn.BatchNorm2d(C).cuda()
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(gam))
        bn.bias.copy_(torch.from_numpy(bet))
    bn.train()
    xd = torch.from_numpy(x).cuda().to(dtype).requires_grad_(True)
    rd = torch.from_numpy(r).cuda().to(dtype).requires_grad_(True) if res else None
    y = HF.bn_act(xd, bn, rd, relu)
    y.backward(torch.from_numpy(gy).cuda().to(dtype))
    # reference in float64 on the SAME (possibly bf16-rounded) inputs
    xr = xd.detach().double().cpu().requires_grad_(True)
    rr = rd.detach().double().cpu().requires_grad_(True) if res else None
    ref = torch.nn.BatchNorm2d(C).double()
    with torch.no_grad():
        ref.weight.copy_(torch.from_numpy(gam))
        ref.bias.copy_(torch.from_numpy(bet))
    ref.train()
    yr = ref(xr)
    if res:
        yr = yr + rr
    if relu:
        yr = torch.relu(yr)
    yr.backward(torch.from_numpy(gy).cuda().to(dtype).double().cpu())
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert torch.allclose(y.detach().double().cpu(), yr.detach(), rtol=tol, atol=tol)
    assert torch.allclose(bn.running_mean.double().cpu(), ref.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn.running_var.double().cpu(), ref.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == 1
    # ReLU masks can differ where y is within rounding of 0; compare gradients away from those pixels
    gx, gxr = xd.grad.double().cpu(), xr.grad
    scale = gxr.abs().max()
    bad = ((gx - gxr).abs() > (1e-4 if dtype == torch.float32 else 3e-2) * scale).float().mean()
    assert bad < (1e-4 if dtype == torch.float32 else 2e-2)
    assert torch.allclose(bn.weight.grad.double().cpu(), ref.weight.grad, rtol=2e-2 if dtype != torch.float32 else 1e-4,
                          atol=2e-2 * float(ref.weight.grad.abs().max()) if dtype != torch.float32 else 1e-4)
    if res:
        assert ((rd.grad.double().cpu() - rr.grad).abs() > 1e-6).float().mean() < 2e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bn_act_eval_vs_torch(K, dtype):
    from hiast_amd import functional as HF
    shape = (2, 32, 9, 13)
    x = synth.normal_f32(111, shape, 2.0)
    r = synth.normal_f32(112, shape, 1.0)
    bn = torch.nn.BatchNorm2d(32).cuda()
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(1 + synth.normal_f32(113, (32,), 0.2)))
        bn.bias.copy_(torch.from_numpy(synth.normal_f32(114, (32,), 0.2)))
        bn.running_mean.copy_(torch.from_numpy(synth.normal_f32(115, (32,), 0.5)))
        bn.running_var.copy_(torch.from_numpy(0.5 + synth.rng(116).random(32, dtype=np.float32)))
    bn.eval()
    xd, rd = torch.from_numpy(x).cuda().to(dtype), torch.from_numpy(r).cuda().to(dtype)
    y = HF.bn_act(xd, bn, rd, True)
    want = torch.relu(bn(xd.float()) + rd.float())
    assert torch.allclose(y.float(), want, rtol=1e-5 if dtype == torch.float32 else 1e-2,
                          atol=1e-5 if dtype == torch.float32 else 2e-2)


def test_trunk_fused_matches_plain_modules(K):
    """whole DeepLab_V2 forward on the device (fused BN/ReLU/add + HIP ASPP) vs the same weights through the
    oracle's functional torch-CPU restatement: eval fp32, logits <= 1e-3 relative (the north_star contract)."""
    from oracle import deeplab_ref
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import SEG_MODEL
    from make_golden import seeded_state_dict
    m = SEG_MODEL["DeepLab_V2"](19, 256)
    sd = seeded_state_dict(m, 9000)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    x = torch.from_numpy(synth.normal_f32(901, (1, 3, 129, 257)))
    with torch.no_grad():
        pred, feat = m(x.cuda())
        want, wfeat = deeplab_ref.deeplab_v2(x, sd)
    err = (pred.cpu() - want).abs().max() / want.abs().max()
    assert err < 1e-3, float(err)
    m.train()
    xb = torch.from_numpy(synth.normal_f32(902, (2, 3, 65, 129)))
    with torch.no_grad():
        pred, _ = m(xb.cuda())
        want, _ = deeplab_ref.deeplab_v2(xb, sd, train=True)
    err = (pred.cpu() - want).abs().max() / want.abs().max()
    assert err < 2e-3, float(err)


def test_bn_act_nhwc_infer(K):
    M, C = 1234, 64
    x = synth.normal_f32(211, (M, C), 2.0)
    bn = torch.nn.BatchNorm2d(C).cuda().eval()
    with torch.no_grad():
        bn.running_mean.copy_(torch.from_numpy(synth.normal_f32(212, (C,), 0.3)))
        bn.running_var.copy_(torch.from_numpy(0.5 + synth.rng(213).random(C, dtype=np.float32)))
        bn.weight.copy_(torch.from_numpy(1 + synth.normal_f32(214, (C,), 0.2)))
    y = K.bn_act_nhwc_infer(dev(x), bn, True)
    want = torch.relu(bn(dev(x).t().reshape(1, C, M, 1))).reshape(C, M).t()
    assert torch.allclose(y, want, rtol=1e-5, atol=1e-5)


def test_teacher_bf16_fast_path_close_to_library_path(K):
    """EMA-teacher forward under bf16 autocast: channels-last fused kernels vs the plain module path"""
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import SEG_MODEL
    from make_golden import seeded_state_dict
    m = SEG_MODEL["DeepLab_V2"](19, 256)
    m.load_state_dict(seeded_state_dict(m, 9000))
    m = m.cuda().eval()
    x = torch.from_numpy(synth.normal_f32(903, (2, 3, 129, 257))).cuda()
    import os
    with torch.no_grad():
        ref, feat = m(x)                                 # fp32-class fast path (split planes)
        assert feat.shape[1] == 2048 and feat.dtype == torch.float32
        with torch.autocast("cuda", dtype=torch.bfloat16):
            fast, featb = m(x)                           # bf16 fast path
            assert featb.dtype == torch.bfloat16 and featb.shape == feat.shape
        os.environ["HIAST_NO_FAST_EVAL"] = "1"
        try:
            slow, feat_slow = m(x)                       # module path: library convs + fused BN kernels, fp32
        finally:
            os.environ.pop("HIAST_NO_FAST_EVAL", None)
    assert (ref - slow).abs().max() <= 1e-3 * slow.abs().max()           # the logits contract
    assert (feat - feat_slow).abs().max() <= 1e-3 * feat_slow.abs().max()
    rel = (fast.float() - ref).abs().max() / ref.abs().max()
    assert rel < 5e-2, float(rel)                        # bf16 through ~100 layers
