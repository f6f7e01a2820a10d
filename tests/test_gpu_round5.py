"""Round-5 additions: the round-4 advisor findings, the captured training step, the 128 x 128 tile form and the early-barrier
k-loop of the tile kernel (igemm_kernel.h).  (The one-launch bottleneck tail K9m — measured slower than the two launches it
replaced — left the default build in round 6: tools/experiments/b2b/ keeps the kernel, its tests and its profile.)"""
import os

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


def _r16(a, dt):
    return torch.from_numpy(a).to(dt).float().numpy()


# ------------------------------------------------------------------------------------------------ round-4 advisor findings
def test_shared_input_gradient_is_not_dropped_without_a_taker():
    """the xsum hand-off of a stage-entry block (conv1 'gives' its data gradient to the downsample convolution's epilogue): a
    'give' convolution whose partner never registered as taker (its leg took another path) must return its input gradient itself
    — round 4 parked it in the shared dict, where nothing picked it up"""
    import torch.nn as nn
    from hiast_amd import functional as HF
    torch.manual_seed(7)
    dev = torch.device("cuda:0")
    conv = nn.Conv2d(512, 256, 1, bias=False).to(dev)
    x0 = torch.randn(2, 512, 24, 40, device=dev).half().contiguous(memory_format=torch.channels_last)
    gy = torch.randn(2, 256, 24, 40, device=dev).half().contiguous(memory_format=torch.channels_last)
    grads = []
    for xsum in (None, ({}, "give")):
        src = x0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.float16):
            y = HF.conv_nhwc(src * 1.0, conv, xsum=xsum)
        y.backward(gy)
        assert src.grad is not None and float(src.grad.abs().max()) > 0
        grads.append(src.grad.clone())
    assert torch.equal(grads[0], grads[1])
    # ... and with a registered taker the pair still delivers ONE summed gradient (tests/test_gpu_round4.py covers the values)
    xs = {}
    src = x0.clone().requires_grad_(True)
    xin = src * 1.0
    conv_t = nn.Conv2d(512, 256, 1, bias=False).to(dev)
    with torch.autocast("cuda", dtype=torch.float16):
        yt = HF.conv_nhwc(xin, conv_t, xsum=(xs, "take"))
        yg = HF.conv_nhwc(xin, conv, xsum=(xs, "give"))
    assert xs.get("taker") is True
    (yg.float() * gy.float()).sum().add((yt.float() * gy.float()).sum()).backward()
    ref = x0.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16):
        r = (HF.conv_nhwc(ref * 1.0, conv).float() * gy.float()).sum() + (HF.conv_nhwc(ref * 1.0, conv_t).float() * gy.float()).sum()
    r.backward()
    d = (src.grad.float() - ref.grad.float()).abs()
    assert (d <= 2.0 ** -9 * ref.grad.float().abs() + 2.0 ** -9 * ref.grad.float().abs().max()).all(), float(d.max())


def test_training_stem_keeps_the_image_gradient():
    """the K9k stem never computes the image gradient: a batch that requires one (adversarial inputs, saliency) must take the
    library stem, as it did before K9k existed"""
    from hiast_amd.sseg.models.modules.resnet import build_resnet101
    torch.manual_seed(3)
    net = build_resnet101(output_stride=8).cuda().train()
    x = torch.randn(1, 3, 64, 128, device="cuda", requires_grad=True)
    with torch.autocast("cuda", dtype=torch.float16):
        y = net(x)
    y.float().square().mean().backward()
    assert x.grad is not None and bool(torch.isfinite(x.grad).all()) and float(x.grad.abs().max()) > 0


# ------------------------------------------------------------------------------------------------ the captured training step
@pytest.mark.parametrize("amp", ["fp16", "bf16"])
def test_graphed_training_step_gradients_bit_equal_to_the_eager_step(tmp_path_factory, monkeypatch, amp):
    """GraphedTrainStep (teacher forward, student forward, fused loss, backward incl. the grouped weight gradients and both
    trunks' weight re-packing as ONE HIP graph) against the eager launches of the same step: every gradient tensor and the four
    losses bit-equal — on the captured batch and on NEW batches pushed through the static buffers; after an optimiser step
    (the re-pack inside the graph picks the new weights up) the replay still equals the eager step"""
    import test_gpu_trainstep_oracle as TS
    from hiast_amd.workflows.trainer.consistency_self_training_trainer import GraphedTrainStep
    depth = "r26"
    TS._patch_depth(monkeypatch, depth)
    root = str(tmp_path_factory.mktemp("graphed_" + amp))
    torch.save(TS._state(depth), os.path.join(root, "init.pth"))
    tr = TS._trainer(root, "O1", amp)
    if tr.scaler is not None:       # a scale at which nothing overflows: NaN would make the comparison vacuous
        tr.scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 8, growth_interval=10 ** 6)
    dev = tr.device
    weak, strong, plbl = (torch.from_numpy(a).to(dev) for a in TS._inputs())
    weak2, strong2 = (weak * 0.9 + 0.05).contiguous(), (strong.flip(3) * 1.1).contiguous()
    plbl2 = plbl.flip(2).contiguous()
    gs = GraphedTrainStep(tr)
    params = dict(tr.model.module.named_parameters())

    def eager(w, s, p):
        tr.g_optimizer.zero_grad(set_to_none=True)
        losses = gs._run(w, s, p)
        torch.cuda.synchronize()
        return ({k: v.detach().clone() for k, v in losses.items()},
                {k: q.grad.detach().clone() for k, q in params.items() if q.grad is not None})

    def graphed(w, s, p):
        losses = gs(w, s, p)
        torch.cuda.synchronize()
        return ({k: v.detach().clone() for k, v in losses.items()},
                {k: q.grad.detach().clone() for k, q in params.items() if q.grad is not None})

    def same(a, b, what):
        assert set(a[0]) == set(b[0]) and set(a[1]) == set(b[1]) and len(a[1]) >= 30
        for k in a[0]:
            assert torch.equal(a[0][k], b[0][k]), (what, k, float(a[0][k]), float(b[0][k]))
        for k in a[1]:
            assert bool(torch.isfinite(a[1][k]).all()), (what, k)
            assert torch.equal(a[1][k], b[1][k]), (what, k, float((a[1][k] - b[1][k]).abs().max()))

    e1, e2 = eager(weak, strong, plbl), eager(weak2, strong2, plbl2)
    assert not torch.equal(e1[1]["seg_model.backbone.conv1.weight"], e2[1]["seg_model.backbone.conv1.weight"])
    for _ in range(GraphedTrainStep.WARM):
        g = graphed(weak, strong, plbl)             # eager iterations of the shape
        assert gs.graph is None
        same(e1, g, "warm-up iteration")
    g = graphed(weak, strong, plbl)                 # capture + first replay
    assert gs.graph is not None
    same(e1, g, "capture")
    same(e2, graphed(weak2, strong2, plbl2), "replay on a new batch")
    same(e1, graphed(weak, strong, plbl), "replay on the first batch again")
    # an optimiser step moves the weights: the graph's own re-pack launches must pick them up
    from hiast_amd.workflows.trainer.consistency_self_training_trainer import StepLosses
    assert isinstance(gs(weak, strong, plbl), StepLosses)     # (the "backward has run" marker travels WITH the losses)
    torch.cuda.synchronize()
    tr.update_model(tr.g_optimizer, tr.d_optimizer, StepLosses(g[0]))
    tr.after_update(1)
    g3 = graphed(weak2, strong2, plbl2)
    gsaved = gs.graph
    gs.graph, gs.key = None, None                   # (eager reference on the updated weights; the static gradients stay allocated)
    e3 = eager(weak2, strong2, plbl2)
    assert not torch.equal(e3[1]["seg_model.backbone.layer3.0.conv2.weight"], e2[1]["seg_model.backbone.layer3.0.conv2.weight"])
    same(e3, g3, "replay after an optimiser step")
    del gsaved


# ------------------------------------------------------------------------------------------------ the 128 x 128 tile form
@pytest.mark.parametrize("shape", [(1, 64, 64, 256, 128), (2, 50, 41, 512, 256), (1, 64, 64, 1024, 256), (1, 72, 64, 128, 512)])
@pytest.mark.parametrize("fmt_name", ["split", "fp16"])
def test_tile_kernel_half_tile_form_equals_the_256_row_form(K, shape, fmt_name, monkeypatch):
    """igemm_kernel.h with BM = 128 (two 4-wave blocks per CU; automatic for 1x1 launches of <= 64 blocks, HIAST_IGEMM_HALF=1 forces
    it): every launch variant the step uses against the 256-row form on the same operands — outputs bit-equal (the same k order
    per element), the per-block statistics equal after their reduction — and against float64 for the fused BN + ReLU form;
    ragged M"""
    B, H, W, Cin, Cout = shape
    assert B * H * W >= 4096
    x = synth.normal_f32(5300, (B, H, W, Cin))
    w = synth.normal_f32(5301, (Cout, Cin, 1, 1), (2.0 / Cin) ** 0.5)
    res = synth.normal_f32(5302, (B, H, W, Cout))
    bn, bnref = _mk_bn(5303, Cout)
    if fmt_name == "split":
        PL = 2
        xp = K.split_planes(dev(x).view(-1, Cin)).view(B, H, W, 2 * Cin)
        rp = K.split_planes(dev(res).view(-1, Cout)).view(B, H, W, 2 * Cout)
        wp = K.pack_conv_weight(dev(w), 2)
        xv, wv = sum(_planes_ref(x)), sum(_planes_ref(w))
    else:
        PL = 1
        xp, rp = dev(x).half(), dev(res).half()
        wp = K.pack_conv_weight(dev(w), K.FMT_FP16)
        xv, wv = _r16(x, torch.float16), _r16(w, torch.float16)

    def run(half):
        monkeypatch.setenv("HIAST_XCONV", "0")              # (keep the expanding shapes on the tile kernel)
        monkeypatch.setenv("HIAST_XCONV2", "0")
        with K.force_half_tile(int(half)):                  # (thread-local override of the library, ABI 6)
            return run_form()

    def run_form():
        out = {"bn_relu": K.igemm_bn_act(xp, wp, PL, bn, None, True), "bn_res_relu": K.igemm_bn_act(xp, wp, PL, bn, rp, True)}
        if PL == 1:
            out["plain"] = K.igemm_bn_act(xp, wp, 1, None, None, False)
            y, part = K.igemm_bn_act(xp, wp, 1, None, None, False, want_stats=True)
            out["stats_y"], out["stats"] = y, K.bn_nhwc_stats_from_partial(part)
            bx = dev(res).half()
            sm, si = dev(0.1 * synth.normal_f32(5304, (Cout,))), dev(np.abs(synth.normal_f32(5305, (Cout,))) + 0.5)
            da, bpart = K.igemm_dgrad_bn_stats(xp, wp, 1, bx, None, None, sm, si)
            out["dgrad"], out["dgrad_sums"] = da, K.bn_nhwc_stats_from_partial(bpart)
            out["rows"] = (part.shape[0], bpart.shape[0])
        torch.cuda.synchronize()
        return out

    full, half = run("0"), run("1")
    M = B * H * W
    if PL == 1:
        assert full["rows"] == ((M + 255) // 256,) * 2 and half["rows"] == ((M + 127) // 128,) * 2
    for k in full:
        if k == "rows":
            continue
        if k in ("stats", "dgrad_sums"):        # another blocking of the same sums
            a, b = full[k].double(), half[k].double()
            assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max())), k
        else:
            assert torch.equal(full[k], half[k]), k
    got = half["bn_relu"]
    got = (K.merge_planes(got.view(-1, 2 * Cout)) if PL == 2 else got.float()).view(B, H, W, Cout).cpu().numpy()
    want = _igemm_ref(xv, wv, bnref, None, True, 1, 1, 1)
    err = np.abs(got - want)
    if PL == 2:
        assert float(err.max()) <= 3e-5 * max(1.0, float(np.abs(want).max()))
    else:
        assert bool((err <= 2.0 ** -10 * np.abs(want) + 3e-5 * np.abs(want).max()).all())


@pytest.mark.parametrize("shape", [(2, 64, 128, 256, 2), (1, 72, 70, 256, 1), (1, 64, 64, 512, 4), (2, 48, 96, 128, 1)])
def test_split_plane_3x3_half_tile_form(K, shape, monkeypatch):
    """the split-plane 3x3 + BN + ReLU launch as 128 x 128 tiles (two blocks per CU; automatic when the 256-row form fills at most a
    quarter of the chip: the generator at batch 2) — with the early-barrier k-loop in a 4-wave block — bit-equal to the 256-row
    form and <= 3e-5 of max against float64; ragged M, dilations"""
    B, H, W, C, dil = shape
    x = np.maximum(synth.normal_f32(5400, (B, H, W, C)), 0)
    w = synth.normal_f32(5401, (C, C, 3, 3), (2.0 / (9 * C)) ** 0.5)
    bn, bnref = _mk_bn(5402, C)
    xp = K.split_planes(dev(x).view(-1, C)).view(B, H, W, 2 * C)
    wp = K.pack_conv_weight(dev(w), 2)
    with K.force_half_tile(0):
        full = K.igemm_bn_act(xp, wp, 2, bn, None, True, 1, dil)
    with K.force_half_tile(1):
        half = K.igemm_bn_act(xp, wp, 2, bn, None, True, 1, dil)
        for _ in range(2):
            assert torch.equal(K.igemm_bn_act(xp, wp, 2, bn, None, True, 1, dil), half)
    assert torch.equal(full, half)
    auto = K.igemm_bn_act(xp, wp, 2, bn, None, True, 1, dil)          # (whichever form the rule picks)
    assert torch.equal(auto, full)
    got = K.merge_planes(half.view(-1, 2 * C)).view(B, H, W, C).cpu().numpy()
    want = _igemm_ref(sum(_planes_ref(x)), sum(_planes_ref(w)), bnref, None, True, 1, dil, 9)
    assert float(np.abs(got - want).max()) <= 3e-5 * max(1.0, float(np.abs(want).max()))


# ------------------------------------------------------------------------------------------------ the early-barrier k-loop
@pytest.mark.parametrize("shape", [(8, 64, 128, 256, 2), (2, 37, 53, 512, 4), (1, 64, 64, 128, 1), (3, 24, 40, 64, 1)])
def test_early_barrier_loop_is_repeatable_and_exact(K, shape):
    """the split-plane 3x3 launches run the early-barrier k-loop (igemm_kernel.h: the barrier of a k-step in the middle of the
    previous one, first fragments prefetched through the fragment rings, two-step B prefetch on two stages): 40 back-to-back
    launches per shape — full-chip, ragged, 128- and 64-column tiles — give one bit pattern (a stage overwritten early or read
    before it landed would not), and that pattern is within 3e-5 of max of float64"""
    B, H, W, C, dil = shape
    x = np.maximum(synth.normal_f32(5500, (B, H, W, C)), 0)
    w = synth.normal_f32(5501, (C, C, 3, 3), (2.0 / (9 * C)) ** 0.5)
    bn, bnref = _mk_bn(5502, C)
    xp = K.split_planes(dev(x).view(-1, C)).view(B, H, W, 2 * C)
    wp = K.pack_conv_weight(dev(w), 2)
    first = K.igemm_bn_act(xp, wp, 2, bn, None, True, 1, dil)
    other = torch.randn(1 << 22, device="cuda")            # (different kernels in between: LDS and caches do not stay warm)
    for i in range(40):
        if i % 8 == 0:
            other = other * 1.0001
        assert torch.equal(K.igemm_bn_act(xp, wp, 2, bn, None, True, 1, dil), first), i
    if B * H * W <= 16384:
        got = K.merge_planes(first.view(-1, 2 * C)).view(B, H, W, C).cpu().numpy()
        want = _igemm_ref(sum(_planes_ref(x)), sum(_planes_ref(w)), bnref, None, True, 1, dil, 9)
        assert float(np.abs(got - want).max()) <= 3e-5 * max(1.0, float(np.abs(want).max()))
