"""Size-independent properties at BASELINE's full sizes (8 images, 1024x512, 2048x64x128 head input), where the
CPU oracle would take minutes: adjoint identities, linearity, conservation of counts, idempotence, determinism."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
B, C, Cin, h, w, H, W = 8, 19, 2048, 64, 128, 512, 1024
DIL = (6, 12, 18, 24)


@pytest.fixture(scope="module")
def K():
    assert torch.cuda.is_available()
    from hiast_amd import kernels
    return kernels


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(*shape, generator=g, device="cuda") * scale


def test_aspp_adjoint_identities_full_size(K):
    """<dy, ASPP(x;W,b)> - <dy, bias> == <dgrad(dy), x> == Σ_i <wgrad_i, W_i>  (bilinearity of the conv)"""
    x = rnd(B, Cin, h, w, seed=1)
    ws = [rnd(C, Cin, 3, 3, seed=2 + i, scale=0.01) for i in range(4)]
    bs = [torch.zeros(C, device="cuda") for _ in range(4)]
    dy = rnd(B, C, h, w, seed=9)
    wpack = K.aspp_pack_weights(ws, bs)
    y = K.aspp_fwd(x, wpack, C, DIL)
    lhs = (y.double() * dy.double()).sum()
    dx = K.aspp_bwd_data(dy, wpack, Cin, DIL)
    mid = (dx.double() * x.double()).sum()
    dws, db = K.aspp_bwd_weight(x, dy, DIL)
    rhs = sum((dw.double() * wt.double()).sum() for dw, wt in zip(dws, ws))
    scale = float((y.double().abs() * dy.double().abs()).sum())
    assert abs(float(lhs - mid)) <= 1e-5 * scale and abs(float(lhs - rhs)) <= 1e-5 * scale
    assert torch.allclose(db, dy.sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
    # determinism (fixed-order split-K reductions, no float atomics)
    assert torch.equal(y, K.aspp_fwd(x, wpack, C, DIL))
    assert torch.equal(dx, K.aspp_bwd_data(dy, wpack, Cin, DIL))
    assert all(torch.equal(a, b_) for a, b_ in zip(dws, K.aspp_bwd_weight(x, dy, DIL)[0]))


def test_upsample_adjoint_and_linearity_full_size(K):
    x, x2 = rnd(B, C, h, w, seed=11), rnd(B, C, h, w, seed=12)
    g = rnd(B, C, H, W, seed=13)
    up = K.upsample_bilinear_ac_fwd(x, H, W)
    gin = K.upsample_bilinear_ac_bwd(g, h, w)
    a, b_ = (up.double() * g.double()).sum(), (x.double() * gin.double()).sum()
    assert abs(float(a - b_)) <= 1e-6 * float((up.double().abs() * g.double().abs()).sum())
    lin = K.upsample_bilinear_ac_fwd(2.0 * x - 3.0 * x2, H, W)
    assert torch.allclose(lin, 2.0 * up - 3.0 * K.upsample_bilinear_ac_fwd(x2, H, W), rtol=1e-5, atol=1e-5)
    # corners are copied exactly (align_corners=True), constants are preserved
    assert torch.equal(up[..., 0, 0], x[..., 0, 0]) and torch.allclose(up[..., -1, -1], x[..., -1, -1], atol=1e-5)
    ones = K.upsample_bilinear_ac_fwd(torch.full((1, 1, h, w), 0.75, device="cuda"), H, W)
    assert float((ones - 0.75).abs().max()) <= 1e-6


def test_pseudo_label_conservation_and_idempotence_full_size(K):
    z = rnd(B, C, h, w, seed=21, scale=3.0)
    mp, am, hist = K.plabel_pass1(z, H, W)
    assert int(hist.long().sum()) == B * H * W                          # every pixel lands in exactly one bin
    per_class = torch.bincount(am.flatten().long(), minlength=C)
    assert torch.equal(hist.long().sum(1), per_class)
    assert float(mp.min()) >= 1.0 / C - 1e-6 and float(mp.max()) <= 1.0
    # consistency with the materialised path: argmax of the upsampled logits
    up = K.upsample_bilinear_ac_fwd(z, H, W)
    assert torch.equal(am.long(), up.argmax(1))
    # no threshold: labels == argmax, counts == bincount, Σprob*2^30 exact
    plbl, count, sfx = K.plabel_pass2(mp, am, None, C)
    assert torch.equal(plbl, am) and torch.equal(count.sum(0), per_class)
    want = torch.zeros(C, dtype=torch.int64, device="cuda").index_add_(
        0, am.flatten().long(), (mp.flatten().double() * 2.0 ** 30).long())
    assert torch.equal(sfx, want)
    # thresholds: monotone (raising every threshold can only remove pixels), 255 exactly where prob < thr[label]
    t1 = torch.full((C,), 0.5, device="cuda")
    t2 = torch.full((C,), 0.8, device="cuda")
    p1, c1, _ = K.plabel_pass2(mp, am, t1, C)
    p2, c2, _ = K.plabel_pass2(mp, am, t2, C)
    assert torch.equal(p1 == 255, mp < 0.5) and torch.equal(p2 == 255, mp < 0.8)
    assert bool((c2 <= c1).all()) and bool(((p2 != 255) <= (p1 != 255)).all())
    # a second launch accumulates into the same histogram (the CBST policy relies on it)
    _, _, hist2 = K.plabel_pass1(z, H, W, hist.clone())
    assert torch.equal(hist2, 2 * hist)


def test_fused_loss_scaling_properties_full_size(K):
    """sums are additive over the batch; the gradient is linear in the upstream coefficients"""
    z, zt = rnd(B, C, h, w, seed=31, scale=2.0), rnd(B, C, h, w, seed=32, scale=2.0)
    g = torch.Generator(device="cuda").manual_seed(33)
    pl = torch.randint(0, C, (B, H, W), generator=g, device="cuda", dtype=torch.uint8)
    pl[torch.rand(B, H, W, generator=g, device="cuda") < 0.4] = 255
    s_all = K.st_loss_fwd(z, zt, pl, H, W, "ignored")
    s_a = K.st_loss_fwd(z[:3].contiguous(), zt[:3].contiguous(), pl[:3].contiguous(), H, W, "ignored")
    s_b = K.st_loss_fwd(z[3:].contiguous(), zt[3:].contiguous(), pl[3:].contiguous(), H, W, "ignored")
    assert torch.equal(s_all[4:7], s_a[4:7] + s_b[4:7])                 # counts: exact
    assert torch.allclose(s_all[:4], s_a[:4] + s_b[:4], rtol=1e-9)
    assert float(s_all[4] + s_all[5]) == B * H * W
    c1 = torch.tensor([1.0, 0.1, 1.0, 0.5], device="cuda")
    d1 = K.st_loss_bwd(z, zt, pl, H, W, "ignored", s_all, c1)
    d2 = K.st_loss_bwd(z, zt, pl, H, W, "ignored", s_all, 2 * c1)
    assert torch.allclose(d2, 2 * d1, rtol=1e-5, atol=1e-12)
    # softmax gradients sum to zero over the class axis at every low-res cell
    assert float(d1.sum(1).abs().max()) <= 1e-6 * float(d1.abs().max()) * C
    # int64 labels give the same result as uint8 labels
    s64 = K.st_loss_fwd(z, zt, pl.long(), H, W, "ignored")
    assert torch.equal(s64, s_all)


def test_ema_idempotence_and_fixed_point(K):
    n = 5_000_000
    e, p = rnd(n, seed=41), rnd(n, seed=42)
    plan = K.EmaPlan([e], [p])
    K.ema_update(plan, 1.0)                      # gamma = 1: unchanged
    assert torch.equal(e, rnd(n, seed=41))
    e.copy_(p)
    K.ema_update(plan, 0.999)                    # ema == p is (nearly) a fixed point
    assert torch.allclose(e, p, rtol=1e-6, atol=0)


def test_unsupported_shapes_are_refused_not_faulted(K):
    from hiast_amd._lib import HiastLibraryError
    with pytest.raises(HiastLibraryError):       # class count the kernels are not built for
        K.plabel_pass1(torch.zeros(1, 5, 4, 4, device="cuda"), 8, 8)
    with pytest.raises(HiastLibraryError):       # down-sampling is not the pseudo-label path
        K.plabel_pass1(torch.zeros(1, 19, 8, 8, device="cuda"), 4, 4)
    with pytest.raises(HiastLibraryError):       # > 253x up-sampling of the fused loss
        K.st_loss_workspace(1, 19, 2, 2, 1024, 2048, torch.device("cuda"))
    with pytest.raises((HiastLibraryError, AssertionError)):   # ASPP input channels must be a multiple of 64
        K.aspp_fwd(torch.zeros(1, 48, 8, 8, device="cuda"), torch.zeros(33 * 48 * 32 + 32, device="cuda"), 19, DIL)
    # empty batch: nothing is launched, shapes are right
    mp, am, hist = K.plabel_pass1(torch.zeros(0, 19, 8, 16, device="cuda"), 64, 128)
    assert mp.shape == (0, 64, 128) and int(hist.sum()) == 0


def test_igemm_trunk_properties_full_size(K):
    """the LDS-DMA convolution kernel at the bench shapes (8 x 64 x 128 pixels): linearity in the input, determinism,
    zero-padding (a one-pixel impulse spreads to exactly the 9 dilated taps), and refusal of shapes it does not tile"""
    from hiast_amd._lib import HiastLibraryError
    Bn, Hh, Ww, Ci, Co, dil = 8, 64, 128, 256, 256, 2
    w = rnd(Co, Ci, 3, 3, seed=21, scale=(2.0 / (9 * Ci)) ** 0.5)
    wp = K.pack_conv_weight(w, 2)
    xa, xb = rnd(Bn * Hh * Ww, Ci, seed=22), rnd(Bn * Hh * Ww, Ci, seed=23)
    pa, pb = (K.split_planes(t).view(Bn, Hh, Ww, 2 * Ci) for t in (xa, xb))
    pab = K.split_planes(xa + xb).view(Bn, Hh, Ww, 2 * Ci)
    ya, yb, yab = (K.merge_planes(K.igemm_bn_act(p, wp, 2, None, None, False, 1, dil).view(-1, 2 * Co)) for p in (pa, pb, pab))
    assert (yab - (ya + yb)).abs().max() <= 5e-5 * yab.abs().max()
    assert torch.equal(ya, K.merge_planes(K.igemm_bn_act(pa, wp, 2, None, None, False, 1, dil).view(-1, 2 * Co)))
    # impulse at a corner pixel of image 3, channel 5: the response is non-zero exactly at the in-image taps
    imp = torch.zeros(Bn, Hh, Ww, Ci, device="cuda")
    imp[3, 0, 0, 5] = 1.0
    yi = K.merge_planes(K.igemm_bn_act(K.split_planes(imp.view(-1, Ci)).view(Bn, Hh, Ww, 2 * Ci), wp, 2, None, None, False,
                                       1, dil).view(-1, 2 * Co)).view(Bn, Hh, Ww, Co)
    nz = (yi.abs().sum(-1) > 0).nonzero().tolist()
    assert sorted(nz) == sorted([[3, yy, xx] for yy in (0, dil) for xx in (0, dil)])
    whl = sum(t.float() for t in K.pack_conv_weight(w, 2).view(Co, 9, Ci // 32, 2, 32).unbind(3)).reshape(Co, 9, Ci)
    assert torch.allclose(yi[3, dil, dil], whl[:, 0, 5], rtol=0, atol=1e-6)      # tap (-1,-1) of w reaches (+d,+d)
    with pytest.raises(HiastLibraryError):       # channel counts the kernel does not tile
        K.igemm_bn_act(torch.zeros(1, 8, 8, 2 * 48, device="cuda", dtype=torch.bfloat16),
                       torch.zeros(64, 1, 2 * 48, device="cuda", dtype=torch.bfloat16), 2, None, None, False)


def test_full_resolution_2048x1024_inference_matches_module_path(K, monkeypatch):
    """BASELINE configs[4] geometry: one 2048x1024 image through the fp32-class fast path (split planes, 128x256 head
    map) against the module path (library convolutions + fused BN kernels, fp32): logits within the 1e-3 contract,
    identical argmax label map up to genuine near-ties"""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import SEG_MODEL
    from make_golden import seeded_state_dict
    m = SEG_MODEL["DeepLab_V2"](19, 256)
    m.load_state_dict(seeded_state_dict(m, 9000))
    m = m.cuda().eval()
    x = rnd(1, 3, 1024, 2048, seed=31)
    with torch.no_grad():
        fast, _ = m(x, need_feat=False)
        monkeypatch.setenv("HIAST_NO_FAST_EVAL", "1")
        slow, _ = m(x)
    assert tuple(fast.shape) == (1, 19, 128, 256)
    assert (fast - slow).abs().max() <= 1e-3 * slow.abs().max()
    top2 = slow.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 2e-3 * slow.abs().max()
    assert torch.equal(fast.argmax(1)[clear], slow.argmax(1)[clear])
