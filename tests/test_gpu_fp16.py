"""The fp16 operand format (HIAST_FMT_FP16 — the reference's apex-O1 type, code/utils/default_config.py:109,
utils/utils.py:126-132) of every kernel of the mixed-precision path, each against float64 on the SAME fp16-rounded
operands (the bf16 variants have the same tests in test_gpu_kernels.py / test_gpu_round2.py; here the tolerance is the
fp16 output rounding 2^-11 instead of bf16's 2^-8), the dynamic-loss-scale handling of the fused Adam step, and the
end-to-end check that `train.amp_dtype: fp16` runs the trunk on the hand-written kernels (no library convolution
besides the 7x7 stem)."""
import numpy as np
import pytest
import torch

import synth
from test_gpu_kernels import IGEMM_CASES, _igemm_ref, _mk_bn, dev

pytestmark = pytest.mark.gpu
H16 = torch.float16


@pytest.fixture(scope="module")
def K():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from hiast_amd import kernels
    return kernels


def _f16r(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _close16(got, want, extra=3e-5):
    """|got - want| <= one fp16 rounding of want + a small share of the largest value (fp32 accumulation order)"""
    return (np.abs(got - want) <= 2.0 ** -10 * np.abs(want) + extra * np.abs(want).max()).all()


@pytest.mark.parametrize("shape", [(64, 64, 1, 1), (256, 128, 3, 3), (128, 320, 1, 1)])
def test_pack_conv_weight_fp16_layout(K, shape):
    """fmt 3: [N][taps][K] fp16 = round-to-nearest-even of the fp32 weight; adjoint = channel-transposed, taps flipped"""
    N, Kc, kh, kw = shape
    w = dev(synth.normal_f32(700, shape, 0.3))
    wp, wpt = K.pack_conv_weight(w, K.FMT_FP16, both=True)
    assert wp.dtype == H16 and tuple(wp.shape) == (N, kh * kw, Kc) and tuple(wpt.shape) == (Kc, kh * kw, N)
    ref = w.reshape(N, Kc, kh * kw).permute(0, 2, 1).to(H16)
    assert torch.equal(wp.view(torch.int16), ref.contiguous().view(torch.int16))
    refa = w.reshape(N, Kc, kh * kw).flip(2).permute(1, 2, 0).to(H16)
    assert torch.equal(wpt.view(torch.int16), refa.contiguous().view(torch.int16))
    assert torch.equal(K.pack_conv_weight(w, K.FMT_FP16, transpose=True).view(torch.int16), wpt.view(torch.int16))


@pytest.mark.parametrize("case", IGEMM_CASES[1:5])
def test_igemm_fp16(K, case):
    """the implicit-GEMM tile kernel on fp16 rows: BN(eval) (+res) + ReLU, plain, fp32 output, statistics epilogue"""
    B, H, W, Cin, Cout, taps, stride, dil, has_res = case
    kk = 3 if taps == 9 else 1
    x = _f16r(synth.normal_f32(720, (B, H, W, Cin)))
    w = synth.normal_f32(721, (Cout, Cin, kk, kk), (2.0 / (Cin * taps)) ** 0.5)
    bn, bnref = _mk_bn(722, Cout)
    xp = dev(x).to(H16)
    wp = K.pack_conv_weight(dev(w), K.FMT_FP16)
    Ho, Wo = (H, W) if taps == 1 else ((H - 1) // stride + 1, (W - 1) // stride + 1)
    res = resp = None
    if has_res:
        res = _f16r(synth.normal_f32(723, (B, Ho, Wo, Cout)))
        resp = dev(res).to(H16)
    for relu in (True, False):
        y = K.igemm_bn_act(xp, wp, 1, bn, resp, relu, stride, dil)
        assert y.dtype == H16 and tuple(y.shape) == (B, Ho, Wo, Cout)
        want = _igemm_ref(x, _f16r(w), bnref, res, relu, stride, dil, taps)
        assert _close16(y.float().cpu().numpy(), want), relu
    if not has_res:
        yf = K.igemm_bn_act(xp, wp, 1, None, None, False, stride, dil, out_f32=True)
        assert yf.dtype == torch.float32
        want = _igemm_ref(x, _f16r(w), None, None, False, stride, dil, taps)
        assert np.abs(yf.cpu().numpy() - want).max() <= 3e-5 * max(1.0, np.abs(want).max())
        y, part = K.igemm_bn_act(xp, wp, 1, None, None, False, stride, dil, want_stats=True)
        yd = y.float().view(-1, Cout).double()
        sums = part.double().sum(0)
        assert torch.allclose(sums[:, 0], yd.sum(0), rtol=1e-5, atol=1e-3) and torch.allclose(sums[:, 1], (yd * yd).sum(0), rtol=1e-5)
    with pytest.raises(TypeError):      # operands of one format only
        K.igemm_bn_act(xp, K.pack_conv_weight(dev(w), K.FMT_BF16), 1, bn, None, True, stride, dil)


@pytest.mark.parametrize("M_hw", [(1, 64, 128), (2, 50, 77)])
def test_xconv_fp16(K, M_hw, monkeypatch):
    """K9e on fp16 rows: every epilogue variant vs float64 on the fp16 operands and vs the tile kernel (HIAST_XCONV=0)"""
    B, H, W = M_hw
    Cin, Cout = 256, 1024
    x = _f16r(synth.normal_f32(740, (B, H, W, Cin)))
    w = synth.normal_f32(741, (Cout, Cin, 1, 1), (2.0 / Cin) ** 0.5)
    res = _f16r(synth.normal_f32(742, (B, H, W, Cout)))
    gate = synth.normal_f32(743, (B, H, W, Cout))
    bits = dev(np.packbits((gate > 0).reshape(B * H * W, Cout // 8, 8), axis=-1, bitorder="little").reshape(B * H * W, Cout // 8))
    bn, bnref = _mk_bn(744, Cout)
    xp, resp = dev(x).to(H16), dev(res).to(H16)
    wp = K.pack_conv_weight(dev(w), K.FMT_FP16)
    cases = [("plain", dict(bn=None, res=None, relu=False), None), ("bn_relu", dict(bn=bn, res=None, relu=True), bnref),
             ("bn_res_relu", dict(bn=bn, res=resp, relu=True), bnref), ("res", dict(bn=None, res=resp, relu=False), None),
             ("gated", dict(bn=None, res=resp, relu=False, res_gate=bits), None)]
    for name, kw, bref in cases:
        args = (xp, wp, 1, kw["bn"], kw["res"], kw["relu"], 1, 1)
        extra = {k: v for k, v in kw.items() if k == "res_gate"}
        y = K.igemm_bn_act(*args, **extra)
        monkeypatch.setenv("HIAST_XCONV", "0")
        y_tile = K.igemm_bn_act(*args, **extra)
        monkeypatch.delenv("HIAST_XCONV")
        rr = None if kw["res"] is None else (res * (gate > 0) if name == "gated" else res)
        want = _igemm_ref(x, _f16r(w), bref, rr, kw["relu"], 1, 1, 1)
        assert _close16(y.float().cpu().numpy(), want), name
        d = (y.float() - y_tile.float()).abs()
        assert float((d > 2.0 ** -10 * y_tile.float().abs() + 1e-5).float().mean()) == 0.0, name
    y, part = K.igemm_bn_act(xp, wp, 1, None, None, False, 1, 1, want_stats=True)
    yd = y.float().view(-1, Cout).double()
    sums = part.double().sum(0)
    assert torch.allclose(sums[:, 0], yd.sum(0), rtol=1e-5, atol=1e-3) and torch.allclose(sums[:, 1], (yd * yd).sum(0), rtol=1e-5)


@pytest.mark.parametrize("shape", [(2, 64, 9, 17), (3, 256, 16, 24), (2, 1024, 5, 7)])
@pytest.mark.parametrize("res,relu", [(False, True), (True, True), (False, False)])
def test_bn_nhwc_fp16_matches_fp64(K, shape, res, relu):
    """training-mode BN (+res)(+ReLU) on channels-last fp16, forward and backward, vs float64 autograd"""
    from hiast_amd import functional as HF
    B, C, H, W = shape
    x = _f16r(synth.normal_f32(760, shape, 2.0) + 0.3)
    r = _f16r(synth.normal_f32(761, shape)) if res else None
    gy = _f16r(synth.normal_f32(762, shape))
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(dev(1.0 + 0.2 * synth.normal_f32(763, (C,))))
        bn.bias.copy_(dev(0.1 * synth.normal_f32(764, (C,))))
    xt = _cl(dev(x).to(H16)).requires_grad_(True)
    rt = _cl(dev(r).to(H16)).requires_grad_(True) if res else None
    y = HF.bn_act(xt, bn, rt, relu)
    assert y.dtype == H16 and y.permute(0, 2, 3, 1).is_contiguous()
    y.backward(_cl(dev(gy).to(H16)))
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
    tol = lambda ref: 2.0 ** -10 * ref.abs() + 3e-4 * ref.abs().max()
    assert ((y.double().cpu() - yd.detach()).abs() <= tol(yd.detach())).all()
    assert ((xt.grad.double().cpu() - xd.grad).abs() <= tol(xd.grad)).all()
    if res:
        assert ((rt.grad.double().cpu() - rd.grad).abs() <= tol(rd.grad)).all()
    n = B * H * W
    assert torch.allclose(bn.running_mean.double().cpu(), 0.1 * mu.detach().flatten(), atol=1e-5)
    assert torch.allclose(bn.running_var.double().cpu(), 0.9 + 0.1 * var.detach().flatten() * n / (n - 1), rtol=1e-5)


@pytest.mark.parametrize("cfg", [(2, 64, 64, 10, 18, 1, 1, 1), (1, 256, 128, 12, 20, 3, 1, 2), (2, 128, 128, 16, 16, 3, 2, 1),
                                 (2, 256, 256, 16, 16, 3, 2, 1), (2, 256, 512, 20, 33, 3, 1, 2), (3, 256, 1024, 33, 21, 1, 1, 1)])
def test_conv_nhwc_autograd_fp16_vs_fp64(K, cfg, monkeypatch):
    """_ConvNhwcFn on fp16: igemm / xconv forward, data gradient on the adjoint weight, own weight gradient (>= 256
    channels; the library's below) vs float64 autograd on the fp16-rounded data; statistics epilogue = a statistics pass"""
    from hiast_amd import functional as HF
    monkeypatch.delenv("HIAST_LIB_WGRAD3", raising=False)
    B, Cin, Cout, H, W, k, stride, dil = cfg
    conv = torch.nn.Conv2d(Cin, Cout, k, stride=stride, padding=dil if k == 3 else 0, dilation=dil if k == 3 else 1,
                           bias=False).cuda()
    with torch.no_grad():
        conv.weight.copy_(dev(synth.normal_f32(780, tuple(conv.weight.shape), (2.0 / (Cin * k * k)) ** 0.5)))
    x = _f16r(synth.normal_f32(781, (B, Cin, H, W)))
    xt = _cl(dev(x).to(H16)).requires_grad_(True)
    assert HF.conv_nhwc_ok(xt, conv)
    with torch.autocast("cuda", dtype=H16):
        y = HF.conv_nhwc(xt, conv)
    assert y.dtype == H16
    gy = _f16r(synth.normal_f32(782, tuple(y.shape)))
    y.backward(_cl(dev(gy).to(H16)))
    xd = torch.from_numpy(x).double().requires_grad_(True)
    wd = torch.from_numpy(_f16r(conv.weight.detach().cpu().numpy())).double().requires_grad_(True)
    yd = torch.nn.functional.conv2d(xd, wd, None, stride, dil if k == 3 else 0, dil if k == 3 else 1)
    yd.backward(torch.from_numpy(gy).double())
    tol = lambda ref: 2.0 ** -10 * ref.abs() + 3e-4 * ref.abs().max()
    # where the library still serves (weight gradients below 256 channels, the data gradient of the strided 3x3) its result
    # arrives as an fp16 tensor accumulated its own way: one more fp16 rounding and a larger share of the maximum
    tol_lib = lambda ref: 2.0 ** -9 * ref.abs() + 2e-3 * ref.abs().max()
    own_w = K.conv_wgrad_preferred(Cin, Cout, k, stride)
    assert ((y.double().cpu() - yd.detach()).abs() <= tol(yd.detach())).all()
    assert ((xt.grad.double().cpu() - xd.grad).abs() <= (tol if stride == 1 or k == 1 else tol_lib)(xd.grad)).all()
    assert conv.weight.grad.dtype == torch.float32
    assert ((conv.weight.grad.double().cpu() - wd.grad).abs() <= (tol if own_w else tol_lib)(wd.grad)).all()
    with torch.autocast("cuda", dtype=H16):
        y2, partial = HF.conv_nhwc(xt.detach(), conv, want_stats=True)
    assert torch.equal(y2, y.detach())
    assert torch.allclose(K.bn_nhwc_stats_from_partial(partial), K.bn_nhwc_stats(y2), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("case", [(2, 24, 40, 1024, 256, 1, 1), (1, 20, 36, 256, 256, 9, 2)])
def test_dgrad_epilogue_bn_backward_sums_fp16(K, case):
    B, H, W, Cdy, Ca, taps, dil = case
    kk = 3 if taps == 9 else 1
    w = dev(synth.normal_f32(800, (Cdy, Ca, kk, kk), (2.0 / (Ca * taps)) ** 0.5))
    wpt = K.pack_conv_weight(w, K.FMT_FP16, transpose=True)
    dy = dev(synth.normal_f32(801, (B, H, W, Cdy))).to(H16)
    x = _cl(dev(synth.normal_f32(802, (B, Ca, H, W), 1.5)).to(H16))
    gamma = dev(synth.normal_f32(803, (Ca,), 0.5)) + 1.0
    beta = dev(synth.normal_f32(804, (Ca,), 0.3))
    xf = x.float()
    sm = xf.mean(dim=(0, 2, 3)).contiguous()
    si = (1.0 / torch.sqrt(xf.var(dim=(0, 2, 3), unbiased=False) + 1e-5)).contiguous()
    plain = K.igemm_bn_act(dy, wpt, 1, None, None, False, 1, dil)
    da, partial = K.igemm_dgrad_bn_stats(dy, wpt, dil, x.permute(0, 2, 3, 1), gamma, beta, sm, si)
    assert da.dtype == H16 and torch.equal(da, plain)
    got = K.bn_nhwc_stats_from_partial(partial).cpu().numpy()
    want = K.bn_nhwc_bwd_stats(da.permute(0, 3, 1, 2), None, x, gamma, beta, sm, si, 2).cpu().numpy()
    assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max() + 1e-6


@pytest.mark.parametrize("shape", [(2, 256, 16, 32, 19), (1, 512, 9, 17, 19)])
def test_aspp2_fp16_fwd_bwd_vs_torch(K, shape):
    """ASPP head on a channels-last fp16 feature: tap GEMM + shift-add forward, gather / dgrad GEMM / transposed-read
    wgrad GEMM backward, vs float64 torch on the fp16-rounded operands"""
    from hiast_amd import functional as HF
    B, Cin, h, w, C = shape
    dil = (6, 12, 18, 24)
    x = _f16r(synth.normal_f32(820, (B, Cin, h, w)))
    ws = [synth.normal_f32(821 + i, (C, Cin, 3, 3), 0.05) for i in range(4)]
    bs = [synth.normal_f32(825 + i, (C,), 0.1) for i in range(4)]
    wt = [dev(t).requires_grad_(True) for t in ws]
    bt = [dev(t).requires_grad_(True) for t in bs]
    xt = _cl(dev(x).to(H16)).requires_grad_(True)
    y = HF.aspp_nhwc(xt, wt, bt, dil)
    assert y.dtype == torch.float32
    gy = synth.normal_f32(830, tuple(y.shape))
    y.backward(dev(gy))
    xd = torch.from_numpy(x).double().requires_grad_(True)
    wd = [torch.from_numpy(_f16r(t)).double().requires_grad_(True) for t in ws]
    bd = [torch.from_numpy(t).double().requires_grad_(True) for t in bs]
    yd = sum(torch.nn.functional.conv2d(xd, wd[i], bd[i], 1, dil[i], dil[i]) for i in range(4))
    yd.backward(torch.from_numpy(_f16r(gy)).double())
    assert np.abs(y.detach().cpu().numpy() - yd.detach().numpy()).max() <= 2e-3 * float(yd.abs().max())
    assert ((xt.grad.double().cpu() - xd.grad).abs() <= 2.0 ** -9 * xd.grad.abs() + 2e-3 * xd.grad.abs().max()).all()
    for i in range(4):
        assert np.abs(wt[i].grad.double().cpu().numpy() - wd[i].grad.numpy()).max() <= 3e-3 * float(wd[i].grad.abs().max()), i
        db_ref = torch.from_numpy(gy).double().sum(dim=(0, 2, 3))      # the bias gradient sums the fp32 dy itself
        assert torch.allclose(bt[i].grad.double().cpu(), db_ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("shape", [(2, 64, 37, 53), (2, 16, 5, 4)])
def test_stem_pooling_and_stem_tail_fp16(K, shape):
    """K18 on fp16: values and input gradient of nn.MaxPool2d(3, 2, 1), bit for bit (ties included); K9f with fp16 input and
    fp16 output = BN(eval) + ReLU + pooling of the fp16 activations"""
    from hiast_amd import functional as HF
    B, C, H, W = shape
    pool = torch.nn.MaxPool2d(3, 2, 1)
    for case in ("smooth", "ties"):
        v = dev(synth.normal_f32(840, shape, 1.0))
        if case == "ties":
            v = torch.round(v * 2) / 2
        x = _cl(v.to(H16))
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ya, yb = HF.maxpool(xa, pool), pool(xb)
        assert ya.dtype == H16 and torch.equal(ya.detach().view(torch.int16), _cl(yb.detach()).view(torch.int16)), case
        g = _cl(dev(synth.normal_f32(841, tuple(yb.shape), 1.0)).to(H16))
        ya.backward(g)
        yb.backward(g)
        assert torch.equal(xa.grad.view(torch.int16), _cl(xb.grad).view(torch.int16)), case
    if C % 8 == 0:
        bn, _ = _mk_bn(845, C)
        x = _cl(dev(synth.normal_f32(846, shape, 1.5)).to(H16))
        got = K.stem_tail(x, bn, K.FMT_FP16)
        a = torch.relu(torch.nn.functional.batch_norm(x.float(), bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps))
        ref = pool(a.to(H16).float()).to(H16)
        Ho, Wo = ref.shape[2:]
        assert got.dtype == H16 and tuple(got.shape) == (B, Ho, Wo, C)
        d = (got.float() - ref.permute(0, 2, 3, 1).float()).abs()
        assert float((d > 2.0 ** -10 * ref.permute(0, 2, 3, 1).float().abs() + 1e-6).float().mean()) == 0.0


def test_fused_adam_handles_the_loss_scale_on_the_device(K):
    """FusedAdam under torch.amp.GradScaler (apex amp.scale_loss, base_trainer.py:129-131): scaled gradients give the update
    torch.optim.Adam makes on the unscaled ones; a step with an inf gradient changes NOTHING (parameters, moments, applied
    step count) and halves the scale; the next good step continues with the right bias correction — all without the host
    reading found_inf"""
    from hiast_amd.utils.utils import FusedAdam
    torch.manual_seed(3)
    shapes = [(64, 32, 3, 3), (19,), (256, 64, 1, 1)]
    p_own = [torch.randn(s, device="cuda").requires_grad_(True) for s in shapes]
    p_ref = [p.detach().clone().requires_grad_(True) for p in p_own]
    own = FusedAdam(p_own, lr=1e-2, betas=(0.9, 0.999), weight_decay=5e-4)
    ref = torch.optim.Adam(p_ref, lr=1e-2, betas=(0.9, 0.999), weight_decay=5e-4)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 10, growth_factor=2.0, backoff_factor=0.5, growth_interval=1000)
    scaler.scale(torch.zeros((), device="cuda"))        # (the scale tensor is created lazily by the first scale() call)
    applied = 0
    for it in range(6):
        gs = [torch.randn(s, device="cuda") for s in shapes]
        overflow = it in (1, 4)
        scale = float(scaler.get_scale())
        for p, g in zip(p_own, gs):
            p.grad = g * scale
            if overflow:
                p.grad.view(-1)[3] = float("inf")
        before = [p.detach().clone() for p in p_own]
        moments = [own.state[p]["exp_avg"].clone() for p in p_own] if applied else None
        scaler.step(own)
        scaler.update()
        if overflow:
            assert all(torch.equal(a, b.detach()) for a, b in zip(before, p_own)), it
            if moments is not None:
                assert all(torch.equal(m, own.state[p]["exp_avg"]) for m, p in zip(moments, p_own)), it
            assert float(scaler.get_scale()) == scale * 0.5
        else:
            applied += 1
            for p, g in zip(p_ref, gs):
                p.grad = g.clone()
            ref.step()
            for a, b in zip(p_own, p_ref):
                assert torch.allclose(a.detach(), b.detach(), rtol=2e-6, atol=2e-7), it
            assert float(scaler.get_scale()) == scale
        assert own.applied_steps() == applied
    sd = own.state_dict()
    assert all(float(st["step"]) == applied for st in sd["state"].values())


def test_fp16_training_step_runs_the_trunk_on_the_own_kernels(K, monkeypatch, tmp_path):
    """`train.amp_dtype: fp16` (apex O1's type): one ConsistencySelfTrainingTrainer step launches NO library convolution
    (the teacher's stem is hiast_stem_eval since round 3, the student's hiast_stem_train_fwd since round 4; no nn.Conv2d
    module runs its own forward, and aten's convolution_backward is never called) — every bottleneck convolution goes through
    hiast_igemm_bn_act / hiast_xconv — and the step produces finite losses and gradients; the same holds for bf16"""
    from test_gpu_trainstep_oracle import _trainer, _state, _inputs, _patch_depth
    _patch_depth(monkeypatch, "r26")
    root = str(tmp_path)
    torch.save(_state("r26"), root + "/init.pth")
    calls = {"conv_fwd": [], "igemm": 0, "conv_bwd": 0}
    orig_fwd = torch.nn.Conv2d.forward
    orig_ig = K.igemm_bn_act
    from torch.utils._python_dispatch import TorchDispatchMode

    class NoLibraryConv(TorchDispatchMode):           # every aten op of the step passes here (autograd's backward included)
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            if "convolution" in str(func):
                calls["conv_bwd"] += 1
            return func(*args, **(kwargs or {}))

    def spy_fwd(self, x):
        calls["conv_fwd"].append(tuple(self.kernel_size))
        return orig_fwd(self, x)

    def spy_ig(*a, **k):
        calls["igemm"] += 1
        return orig_ig(*a, **k)
    monkeypatch.setattr(torch.nn.Conv2d, "forward", spy_fwd)
    monkeypatch.setattr(K, "igemm_bn_act", spy_ig)
    weak, strong, plbl = _inputs()
    for amp in ("fp16", "bf16"):
        calls["conv_fwd"].clear()
        calls["igemm"] = 0
        tr = _trainer(root, "O1", amp)
        calls["conv_bwd"] = 0
        with NoLibraryConv():
            losses = tr.train_on(dev(weak), dev(strong), dev(plbl))
            tr.update_model(tr.g_optimizer, tr.d_optimizer, losses)
        torch.cuda.synchronize()
        assert calls["conv_fwd"] == [] and calls["conv_bwd"] == 0, (amp, calls["conv_fwd"], calls["conv_bwd"])
        assert calls["igemm"] >= 2 * 28, (amp, calls["igemm"])          # 28 trunk convolutions per forward (+ data gradients)
        assert all(np.isfinite(float(v)) for v in losses.values()), (amp, losses)
        if amp == "fp16":
            # apex's dynamic loss scale: starts at 2^16, every overflow halves it and skips the update; within a few
            # iterations updates are applied (device-side decision: FusedAdam.applied_steps counts them)
            p0 = next(tr.model.module.seg_model.aspp.parameters()).detach().clone()
            for _ in range(15):
                losses = tr.train_on(dev(weak), dev(strong), dev(plbl))
                tr.update_model(tr.g_optimizer, tr.d_optimizer, losses)
            applied, scale = tr.g_optimizer.applied_steps(), float(tr.scaler.get_scale())
            assert 1 <= applied <= 16 and scale <= 2.0 ** 16 and scale == 2.0 ** 16 / 2 ** (16 - applied), (applied, scale)
            assert not torch.equal(p0, next(tr.model.module.seg_model.aspp.parameters()).detach()), "student did not move"
            assert all(np.isfinite(float(v)) for v in losses.values()), losses
