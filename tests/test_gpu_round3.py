"""Round-3 kernels: K9g (xconv2.hip) — the split-plane 256 -> N 1x1 launches of the pseudo-label forward with
register-resident weights — against float64 on the operand values the kernel multiplies and against the tile kernel
(HIAST_XCONV2=0) on the same inputs; K9e (xconv.hip) after its loop barrier became a bare s_barrier behind a counted wait."""
import numpy as np
import pytest
import torch

import synth
from test_gpu_kernels import _igemm_ref, _mk_bn, _planes_ref, dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from hiast_amd import kernels
    return kernels


@pytest.mark.parametrize("M_hw", [(1, 64, 128), (2, 50, 77), (1, 64, 65), (8, 64, 128)])
@pytest.mark.parametrize("Cout", [1024, 256])
def test_xconv2_split_plane_expanding_1x1(K, M_hw, Cout, monkeypatch):
    """conv3 of a layer3 bottleneck in the fp32-class forward (256 -> 1024, + BN(eval) + identity + ReLU): every variant of
    the register-resident-weight kernel vs float64 and vs the tile kernel; ragged M (a tail panel with rows beyond M),
    many panels per block (B = 8: 16 per stream), the stored planes are a valid split (hi + lo re-splits to itself)"""
    B, H, W = M_hw
    Cin = 256
    x = synth.normal_f32(3100, (B, H, W, Cin))
    w = synth.normal_f32(3101, (Cout, Cin, 1, 1), (2.0 / Cin) ** 0.5)
    res = synth.normal_f32(3102, (B, H, W, Cout))
    bn, bnref = _mk_bn(3103, Cout)
    xp = K.split_planes(dev(x).view(-1, Cin)).view(B, H, W, 2 * Cin)
    wp = K.pack_conv_weight(dev(w), 2)
    resp = K.split_planes(dev(res).view(-1, Cout)).view(B, H, W, 2 * Cout)
    xh, xl = _planes_ref(x)
    wh, wl = _planes_ref(w)
    rr = sum(_planes_ref(res))
    lib = K._lib.load()
    for name, r_dev, r_ref, relu in (("bn_res_relu", resp, rr, True), ("bn_relu", None, None, True), ("bn", None, None, False)):
        guard = torch.full((64,), 7, dtype=torch.int16, device="cuda")        # canary behind the output (tail rows)
        y = K.igemm_bn_act(xp, wp, 2, bn, r_dev, relu)
        monkeypatch.setenv("HIAST_XCONV2", "0")
        y_tile = K.igemm_bn_act(xp, wp, 2, bn, r_dev, relu)
        monkeypatch.delenv("HIAST_XCONV2")
        assert bool((guard == 7).all())
        got = K.merge_planes(y.view(-1, 2 * Cout)).view(B, H, W, Cout).cpu().numpy()
        want = _igemm_ref(xh + xl, wh + wl, bnref, r_ref, relu, 1, 1, 1)
        tol = 3e-5 * max(1.0, np.abs(want).max())
        assert np.abs(got - want).max() <= tol, (name, np.abs(got - want).max(), tol)
        tile = K.merge_planes(y_tile.view(-1, 2 * Cout)).view(B, H, W, Cout).cpu().numpy()
        assert np.abs(got - tile).max() <= 2e-5 * max(1.0, np.abs(want).max()), name      # another summation order only
        v = K.merge_planes(y.view(-1, 2 * Cout))
        assert torch.equal(K.merge_planes(K.split_planes(v)), v), name
    # the shape really takes the new kernel (M >= 4096) — and smaller maps stay on the tile kernel
    assert (B * H * W >= 4096)


def test_xconv_variants_after_the_bare_barrier(K, monkeypatch):
    """K9e with the counted wait + bare s_barrier in its panel loop: many panels per block (B = 8), both 16-bit types,
    every epilogue variant vs the tile kernel on the same inputs (the unit tests of the variants vs float64 are
    test_xconv_expanding_1x1 / test_xconv_fp16)"""
    B, H, W, Cin, Cout = 8, 64, 128, 256, 1024
    x = synth.normal_f32(3200, (B, H, W, Cin))
    w = synth.normal_f32(3201, (Cout, Cin, 1, 1), (2.0 / Cin) ** 0.5)
    res = synth.normal_f32(3202, (B, H, W, Cout))
    bits = (torch.rand(B * H * W, Cout // 8, device="cuda") * 256).to(torch.uint8)
    bn, _ = _mk_bn(3203, Cout)
    for dt, fmt in ((torch.float16, K.FMT_FP16), (torch.bfloat16, K.FMT_BF16)):
        xp, resp = dev(x).to(dt), dev(res).to(dt)
        wp = K.pack_conv_weight(dev(w), fmt)
        cases = [dict(bn=None, res=None, relu=False), dict(bn=bn, res=resp, relu=True), dict(bn=bn, res=None, relu=True),
                 dict(bn=None, res=resp, relu=False), dict(bn=None, res=resp, relu=False, res_gate=bits)]
        for kw in cases:
            args = (xp, wp, 1, kw["bn"], kw["res"], kw["relu"], 1, 1)
            extra = {k: v for k, v in kw.items() if k == "res_gate"}
            for rep in range(3):        # (a race between panels would not repeat identically)
                y = K.igemm_bn_act(*args, **extra)
                if rep == 0:
                    first = y.clone()
                assert torch.equal(y, first), (dt, kw.keys(), rep)
            monkeypatch.setenv("HIAST_XCONV", "0")
            y_tile = K.igemm_bn_act(*args, **extra)
            monkeypatch.delenv("HIAST_XCONV")
            d = (y.float() - y_tile.float()).abs()
            ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
            assert float((d > ulp * y_tile.float().abs() + 1e-5).float().mean()) == 0.0, (dt, list(kw))
        y, part = K.igemm_bn_act(xp, wp, 1, None, None, False, 1, 1, want_stats=True)
        yd = y.float().view(-1, Cout).double()
        sums = part.double().sum(0)
        assert torch.allclose(sums[:, 0], yd.sum(0), rtol=1e-5, atol=1e-2) and torch.allclose(sums[:, 1], (yd * yd).sum(0), rtol=1e-5)


@pytest.mark.parametrize("cfg", [(3, 256, 256, 9, 12, 3, 2, 1),      # narrow map: the pixel position is decoded every step
                                 (2, 256, 256, 33, 24, 3, 1, 2),     # stride 2, Wo = 12: decoded, input pixel != output pixel
                                 (2, 256, 512, 31, 48, 3, 2, 2),     # stride 2, Wo = 24: the walked form with row / image jumps
                                 (5, 512, 256, 7, 20, 3, 3, 1),      # stride 1, Wo = 20: tap shift in the descriptor, a 20-pixel
                                                                     # move crosses a whole row; taps reach 3 rows out of 7
                                 (2, 256, 256, 64, 128, 3, 12, 1),   # dilation 12 (the shift is 12 rows + 12 pixels)
                                 (1, 256, 256, 21, 37, 3, 1, 1)])    # odd sizes: ragged last range, lanes past the last pixel
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_conv_wgrad_pixel_walk_forms(K, cfg, dt):
    """K9d after round 3's k-loop (ring of four 32-pixel stages) and 3x3 addressing (stride 1: tap shift in the buffer
    descriptor + in-image test per lane; otherwise a walked or decoded input pixel) against the fp32 weight gradient of the
    same 16-bit operands, every form, both types; bitwise repeatable"""
    B, Cin, Cout, H, W, k, dil, stride = cfg
    x = dev(synth.normal_f32(3300, (B, H, W, Cin))).to(dt)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    dy = dev(synth.normal_f32(3301, (B, Ho, Wo, Cout))).to(dt)
    dw = K.conv_wgrad_nhwc(dy, x, k, stride, dil)
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).float(), (Cout, Cin, k, k), dy.permute(0, 3, 1, 2).float(),
                                      stride=stride, padding=dil, dilation=dil)
    err = (dw - ref).abs().max().item()
    assert err <= 1e-4 * ref.abs().max().item(), (cfg, err, ref.abs().max().item())
    assert torch.equal(dw, K.conv_wgrad_nhwc(dy, x, k, stride, dil))


@pytest.mark.parametrize("C", [19, 16, 9, 2])
def test_plabel_pass1_scattered_workspace_equals_direct_counting(K, C):
    """K3 with the scattered counting workspace (round 3) vs counting straight into the histogram (workspace = NULL): same
    maps, same integer histogram, on a peaked class distribution (two dominant classes, spatially smooth confidences: the
    case the workspace exists for); the histogram accumulates across calls and the workspace is re-zeroed by every call"""
    import ctypes
    g = torch.Generator(device="cuda").manual_seed(3400 + C)
    B, h, w, H, W = 3, 24, 40, 187, 317
    z = torch.nn.functional.interpolate(torch.randn(B, C, 5, 7, device="cuda", generator=g) * 5.0, size=(h, w), mode="bilinear",
                                        align_corners=True).contiguous()
    z[:, 0] += 4.0
    z[:, C - 1] += 3.0
    mp, am, hist = K.plabel_pass1(z, H, W)
    lib = K._lib.load()
    mp0 = torch.empty_like(mp)
    am0 = torch.empty_like(am)
    hist0 = torch.zeros_like(hist)
    rc = lib.hiast_plabel_pass1(ctypes.c_void_p(z.data_ptr()), B, C, h, w, H, W, ctypes.c_void_p(mp0.data_ptr()),
                                ctypes.c_void_p(am0.data_ptr()), ctypes.c_void_p(hist0.data_ptr()), None, 0,
                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    assert torch.equal(mp, mp0) and torch.equal(am, am0) and torch.equal(hist, hist0)
    assert int(hist.sum()) == B * H * W
    _, _, hist2 = K.plabel_pass1(z, H, W, hist.clone())
    assert torch.equal(hist2, 2 * hist)
    # a workspace that is too small is refused, not overrun
    small = torch.empty(16, dtype=torch.int32, device="cuda")
    rc = lib.hiast_plabel_pass1(ctypes.c_void_p(z.data_ptr()), B, C, h, w, H, W, ctypes.c_void_p(mp0.data_ptr()),
                                ctypes.c_void_p(am0.data_ptr()), ctypes.c_void_p(hist0.data_ptr()), ctypes.c_void_p(small.data_ptr()),
                                64, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == -3


SMALL_WGRAD = [(2, 64, 64, 19, 33, 1, 1, 1),      # layer1.0.conv1: both operands 64 wide (TI = 64, half a j tile)
               (2, 256, 64, 19, 33, 1, 1, 1),     # layer1.1.conv1: dY 64 wide, X in two 128-channel windows
               (2, 64, 256, 19, 33, 1, 1, 1),     # layer1.x.conv3 / downsample: computed transposed
               (3, 64, 64, 17, 29, 3, 1, 1),      # layer1.x.conv2: tap pairs side by side in the j tile
               (2, 512, 128, 13, 21, 1, 1, 1),    # layer2.x.conv1
               (2, 256, 128, 26, 42, 1, 1, 2),    # layer2.0.conv1: strided 1x1 (subsampled input)
               (2, 128, 512, 13, 21, 1, 1, 1),    # layer2.x.conv3
               (2, 256, 512, 26, 42, 1, 1, 2),    # layer2.0.downsample
               (3, 128, 128, 13, 21, 3, 1, 1),    # layer2.x.conv2
               (1, 128, 128, 40, 64, 3, 2, 1),    # dilated 3x3, many pixel ranges
               (8, 64, 64, 64, 128, 3, 1, 1),     # 65536 pixels: hundreds of ranges, XCD order with a ragged block count
               (2, 128, 128, 27, 45, 3, 1, 2),    # layer2.0.conv2: strided 3x3 (input pixel from the output position)
               (3, 64, 64, 16, 20, 3, 2, 2)]      # strided + dilated, tap pairs, several images


@pytest.mark.parametrize("cfg", SMALL_WGRAD)
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_conv_wgrad_small_channels(K, cfg, dt):
    """K9h (wgrad_small.hip): weight gradient of the layer1 / layer2 convolutions (64 .. 512 channels) vs the fp32 weight
    gradient of the same 16-bit operands — every tiling form (TI = 64 / 128, 128-channel windows, tap pairs, transposed,
    strided 1x1 on the subsampled input), both types; bitwise repeatable"""
    B, Cin, Cout, H, W, k, dil, stride = cfg
    assert K.conv_wgrad_small_supported(Cin, Cout, k, stride)
    x = dev(synth.normal_f32(3500, (B, H, W, Cin))).to(dt)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    dy = dev(synth.normal_f32(3501, (B, Ho, Wo, Cout))).to(dt)
    dw = K.conv_wgrad_small_nhwc(dy, x, k, stride, dil)
    assert dw.dtype == torch.float32 and tuple(dw.shape) == (Cout, Cin, k, k)
    pad = dil if k == 3 else 0
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).float(), (Cout, Cin, k, k), dy.permute(0, 3, 1, 2).float(),
                                      stride=stride, padding=pad, dilation=dil if k == 3 else 1)
    err = (dw - ref).abs().max().item()
    assert err <= 1e-4 * ref.abs().max().item(), (cfg, err, ref.abs().max().item())
    assert torch.equal(dw, K.conv_wgrad_small_nhwc(dy, x, k, stride, dil))


@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 67, 101), (2, 9, 14), (3, 130, 258), (1, 512, 1024)])
@pytest.mark.parametrize("fmt", [1, 2, 3])
def test_stem_eval_fused(K, shape, fmt):
    """K9j (stem.hip): conv 7x7 s2 -> bn (eval) -> ReLU -> MaxPool2d(3, 2, 1) in one kernel vs float64 torch on the operand
    values the kernel multiplies (16-bit formats: x and W rounded to the type; split planes: the fp32 values), every output
    format, odd sizes (ragged tiles, pooling windows over the map's border), image sizes below one tile"""
    B, H, W = shape
    x = dev(synth.normal_f32(3700, (B, 3, H, W)))
    w = dev(synth.normal_f32(3701, (64, 3, 7, 7), 0.15))
    bn = torch.nn.BatchNorm2d(64).cuda().eval()
    with torch.no_grad():
        bn.weight.copy_(dev(synth.normal_f32(3702, (64,), 0.3)) + 1.0)
        bn.bias.copy_(dev(synth.normal_f32(3703, (64,), 0.3)))
        bn.running_mean.copy_(dev(synth.normal_f32(3704, (64,), 0.5)))
        bn.running_var.copy_(dev(synth.normal_f32(3705, (64,), 0.2)).abs() + 0.5)
    out = K.stem_eval(x, w, bn, fmt)
    dt = {1: torch.bfloat16, 3: torch.float16}.get(fmt)
    xq = x.to(dt).double() if dt is not None else x.double()
    wq = w.to(dt).double() if dt is not None else w.double()
    y = torch.nn.functional.conv2d(xq, wq, None, 2, 3)
    y = torch.nn.functional.batch_norm(y, bn.running_mean.double(), bn.running_var.double(), bn.weight.double(), bn.bias.double(),
                                       False, 0.0, bn.eps).clamp_min(0)
    ref = torch.nn.functional.max_pool2d(y, 3, 2, 1).permute(0, 2, 3, 1)
    Hp, Wp = ref.shape[1], ref.shape[2]
    assert tuple(out.shape) == (B, Hp, Wp, 64 * (2 if fmt == 2 else 1))
    got = (K.merge_planes(out.view(-1, 128)).view(B, Hp, Wp, 64) if fmt == 2 else out.float()).double()
    ulp = {1: 2.0 ** -8, 2: 0.0, 3: 2.0 ** -11}[fmt]
    tol = ulp * ref.abs() + 3e-5 * float(ref.abs().max())
    assert bool(((got - ref).abs() <= tol).all()), (shape, fmt, float((got - ref).abs().max()), float(ref.abs().max()))


def test_stem_eval_in_the_eval_forward_matches_the_module_path(K, monkeypatch):
    """ResNet.forward_eval_planes with the fused stem vs the same forward with HIAST_NO_STEM_FUSED=1 (library convolution +
    hiast_stem_tail): the trunk feature agrees to the rounding of the 16-bit stem output / 1e-4 in the fp32-class format"""
    from hiast_amd.sseg.models.modules.resnet import build_resnet101
    from hiast_amd.tools import synth_data
    torch.manual_seed(5)
    x = dev(synth.normal_f32(3710, (2, 3, 64, 96)))
    m = synth_data.calibrate_bn(build_resnet101(False, 8).cuda(), x).eval()    # (running statistics of the data: fp16 range)
    with torch.no_grad():
        a = m(x)
        monkeypatch.setenv("HIAST_NO_STEM_FUSED", "1")
        b = m(x)
        monkeypatch.setenv("HIAST_NO_FAST_EVAL", "1")
        c = m(x)                                                   # module path: the library's fp32 convolutions
        monkeypatch.delenv("HIAST_NO_STEM_FUSED")
        monkeypatch.delenv("HIAST_NO_FAST_EVAL")
        s = float(c.abs().max())
        # (33 blocks amplify a 1e-5 difference at the stem a few hundred times: the yardstick is how far the tail path is
        # from the library's fp32 forward)
        ea, eb = float((a - c).abs().max()), float((b - c).abs().max())
        assert ea <= 2.0 * eb + 1e-4 * s and ea <= 2e-2 * s, (ea / s, eb / s)
        with torch.autocast("cuda", dtype=torch.float16):
            a16 = m(x).float()
            monkeypatch.setenv("HIAST_NO_STEM_FUSED", "1")
            b16 = m(x).float()
        assert bool(torch.isfinite(a16).all()) and bool(torch.isfinite(b16).all())
        # the fused stem skips one fp16 rounding (the convolution output): at least as close to the fp32 forward
        ea, eb = float((a16 - c).abs().max()), float((b16 - c).abs().max())
        assert ea <= 1.5 * eb + 1e-3 * s, (ea / s, eb / s)

