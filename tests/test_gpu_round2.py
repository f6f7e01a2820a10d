"""Round-2 regression tests of the host side of the HIP path."""
import os

import numpy as np
import pytest
import torch

from hiast_amd import switches as SW

import synth

pytestmark = pytest.mark.gpu


def _model(seed=780):
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import MODEL
    from hiast_amd.utils.default_config import get_default_cfg
    from make_golden import seeded_state_dict
    cfg = get_default_cfg()
    cfg.model.type = "SelfTrainingSegmentor"
    m = MODEL["SelfTrainingSegmentor"](cfg)
    m.load_state_dict({"seg_model." + k: v for k, v in seeded_state_dict(m.seg_model, seed).items()})
    return cfg, m.cuda()


def test_packed_trunk_weights_follow_adam_and_ema():
    """FusedAdam.step and EmaUpdater write parameters through raw pointers; the kernel-format (bf16 / split-plane)
    copies of the trunk weights are cached on Parameter._version and must be re-packed afterwards — the student's
    next forward, the EMA teacher's next forward and the pseudo-label forward all run on the NEW weights."""
    from hiast_amd import kernels as K
    from hiast_amd.utils import utils
    cfg, student = _model()
    _, teacher = _model()
    for p in teacher.parameters():
        p.requires_grad = False
    utils.freeze_bn(student)
    opt = utils.FusedAdam([p for p in student.parameters() if p.requires_grad], lr=1e-2)
    x = torch.from_numpy(synth.normal_f32(41, (2, 3, 64, 128))).cuda()
    plbl = torch.from_numpy(synth.pseudo_labels(42, 2, 64, 128, 19)).cuda()
    conv = student.seg_model.backbone.layer3[4].conv2
    tconv = teacher.seg_model.backbone.layer3[4].conv2

    def student_step():
        student.train()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = student(x, lowres=True)
        loss = sum(student.compute_loss_lowres(out["logits_lowres"], plbl, out["size"]).values())
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    def teacher_eval(autocast):
        teacher.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            return teacher(x, lowres=True)["logits_lowres"].float().clone()

    student_step()
    w_after_1 = conv.weight.detach().clone()
    v1 = conv.weight._version
    student_step()          # this forward must have run on the weights of step 1
    assert conv.weight._version > v1, "the optimiser step must move Parameter._version"
    ent = conv.__dict__["_hiast_packed"][1]
    assert torch.equal(ent[2], K.pack_conv_weight(w_after_1, 1)), "student forward ran on stale packed weights"
    adj = conv.__dict__["_hiast_packed_adj"][1]        # (keyed by operand format: 1 = bf16)
    assert torch.equal(adj[2], K.pack_conv_weight(w_after_1, 1, transpose=True)), "stale adjoint (data-gradient) weights"

    y16_0, y32_0 = teacher_eval(True), teacher_eval(False)
    ema = utils.EmaUpdater()
    ema(teacher, student, 0.5)
    y16_1, y32_1 = teacher_eval(True), teacher_eval(False)
    for PL in (1, 2):
        ent = tconv.__dict__["_hiast_packed"][PL]
        assert torch.equal(ent[2], K.pack_conv_weight(tconv.weight.detach(), PL)), "teacher PL=%d forward on stale weights" % PL
    assert not torch.equal(y16_0, y16_1) and not torch.equal(y32_0, y32_1), "EMA update did not reach the fast eval path"
    # and the fast path agrees with the module path (plain torch convolutions on the live parameters)
    os.environ["HIAST_NO_FAST_EVAL"] = "1"
    try:
        ref = teacher_eval(False)
    finally:
        del os.environ["HIAST_NO_FAST_EVAL"]
    assert float((y32_1 - ref).abs().max()) <= 1e-3 * float(ref.abs().max())


@pytest.mark.parametrize("case", [(2, 24, 40, 1024, 256, 1, 1), (1, 20, 36, 256, 256, 9, 2), (2, 9, 13, 128, 64, 9, 1),
                                  (1, 16, 16, 512, 128, 1, 1)])
def test_dgrad_epilogue_delivers_bn_backward_sums(case):
    """hiast_igemm_dgrad_bn_stats: the data gradient is bit-identical to the plain launch, and the per-block sums reduce
    to what hiast_bn_nhwc_bwd_stats (gate recomputed from x) computes from the same dA and x (fp32 partial sums in another
    order: 1e-4 of the largest sum); nullable gamma / beta; tail rows of the 256-row tile"""
    from hiast_amd import kernels as K
    B, H, W, Cdy, Ca, taps, dil = case          # conv forward: Ca -> Cdy channels; its data gradient: Cdy -> Ca
    kk = 3 if taps == 9 else 1
    dev = torch.device("cuda")
    w = torch.from_numpy(synth.normal_f32(910, (Cdy, Ca, kk, kk), (2.0 / (Ca * taps)) ** 0.5)).to(dev)
    wpt = K.pack_conv_weight(w, 1, transpose=True)
    dy = torch.from_numpy(synth.normal_f32(911, (B, H, W, Cdy))).to(dev).bfloat16()
    x = torch.from_numpy(synth.normal_f32(912, (B, Ca, H, W), 1.5)).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    gamma = torch.from_numpy(synth.normal_f32(913, (Ca,), 0.5)).to(dev) + 1.0
    beta = torch.from_numpy(synth.normal_f32(914, (Ca,), 0.3)).to(dev)
    xf = x.float()
    sm = xf.mean(dim=(0, 2, 3)).contiguous()
    si = (1.0 / torch.sqrt(xf.var(dim=(0, 2, 3), unbiased=False) + 1e-5)).contiguous()
    plain = K.igemm_bn_act(dy, wpt, 1, None, None, False, 1, dil)
    for g, b in ((gamma, beta), (None, None)):
        da, partial = K.igemm_dgrad_bn_stats(dy, wpt, dil, x.permute(0, 2, 3, 1), g, b, sm, si)
        assert torch.equal(da, plain)
        got = K.bn_nhwc_stats_from_partial(partial).cpu().numpy()
        want = K.bn_nhwc_bwd_stats(da.permute(0, 3, 1, 2), None, x, g, b, sm, si, 2).cpu().numpy()
        assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max() + 1e-6, np.abs(got - want).max()
        # and against fp64 on the stored values
        d64 = da.double().permute(0, 3, 1, 2)
        xh = (x.double() - sm.double().view(1, -1, 1, 1)) * si.double().view(1, -1, 1, 1)
        open_ = (x.float() * ((g if g is not None else torch.ones_like(sm)) * si).view(1, -1, 1, 1)
                 + ((b if b is not None else torch.zeros_like(sm)) - sm * (g if g is not None else torch.ones_like(sm)) * si).view(1, -1, 1, 1)) > 0
        gg = torch.where(open_, d64, torch.zeros_like(d64))
        ref = torch.stack([gg.sum(dim=(0, 2, 3)), (gg * xh).sum(dim=(0, 2, 3))], dim=1).cpu().numpy()
        big = np.abs(ref).max()
        # (the gate is an fp32 fmaf on the device: elements within an ulp of zero may differ from this reference's gate)
        assert np.abs(got - ref).max() <= 2e-3 * big, (np.abs(got - ref).max(), big)


def test_bottleneck_backward_with_and_without_statistics_fusion(monkeypatch):
    """a training-mode bottleneck on channels-last bf16: input gradient and weight gradients with the BatchNorm backward
    sums taken from the data-gradient epilogues equal the run that computes them in a pass of their own"""
    from hiast_amd.sseg.models.modules.resnet import Bottleneck
    torch.manual_seed(5)
    blk = Bottleneck(1024, 256, stride=1, dilation=2, downsample=None).cuda().train()
    x0 = torch.from_numpy(synth.normal_f32(920, (2, 1024, 24, 40))).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
    gout = torch.from_numpy(synth.normal_f32(921, (2, 1024, 24, 40))).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setitem(SW.SWITCHES, "HIAST_NO_BN_BWD_FUSION", flag == "1")
        for p in blk.parameters():
            p.grad = None
        x = x0.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = blk(x)
        y.backward(gout)
        from hiast_amd import functional as HF
        HF.wgrad_stream_join()
        torch.cuda.synchronize()
        res[flag] = (x.grad.float().cpu().numpy(), {n: p.grad.float().cpu().numpy() for n, p in blk.named_parameters() if p.grad is not None})
    gx0, gw0 = res["0"]
    gx1, gw1 = res["1"]
    assert np.abs(gx0 - gx1).max() <= 2e-2 * np.abs(gx1).max()
    assert float((gx0 * gx1).sum() / np.sqrt((gx0 ** 2).sum() * (gx1 ** 2).sum())) >= 0.9999
    for n in gw1:
        c = float((gw0[n] * gw1[n]).sum() / np.sqrt((gw0[n] ** 2).sum() * (gw1[n] ** 2).sum() + 1e-30))
        assert c >= 0.9995, (n, c)


def test_eval_forward_in_sub_batches(monkeypatch):
    """HF.eval_forward_split: the fp32-class inference forward of 8 images as two / four sub-batches on their own streams
    equals one forward up to the library's batch-dependent choice of the stem convolution's algorithm (3e-5 of max on this
    random-init net; argmax equal) — also right after the weights changed (the packed copies are refreshed before the
    streams fork: a stale or half-written copy would be off by far more than rounding)"""
    from hiast_amd import functional as HF
    cfg, net = _model(781)
    net.eval()
    x = torch.from_numpy(synth.normal_f32(930, (8, 3, 64, 128))).cuda()

    def close(a, b):
        return bool((a - b).abs().max() <= 2e-4 * b.abs().max()) and float((a.argmax(1) == b.argmax(1)).float().mean()) >= 0.999

    with torch.no_grad():
        ref = net(x, lowres=True)["logits_lowres"].clone()
        for parts in (2, 4):
            out = HF.eval_forward_split(net, x, parts)["logits_lowres"]
            torch.cuda.synchronize()
            assert close(out, ref)
        # weights move (as after an EMA update): every sub-batch must see the new packed copies
        for p in net.parameters():
            p.data.mul_(1.01)
        torch.autograd.graph.increment_version(list(net.parameters()))
        out = HF.eval_forward_split(net, x, 2)["logits_lowres"].clone()
        ref2 = net(x, lowres=True)["logits_lowres"]
        torch.cuda.synchronize()
        assert close(out, ref2) and not close(ref2, ref)


@pytest.mark.parametrize("amp", [None, torch.bfloat16])
def test_graphed_inference_forward_replays_the_eager_launches(amp, monkeypatch):
    """HF.GraphedEval: two eager calls, then the forward is captured into a HIP graph and replayed.  The graph holds the
    launches of the eager forward, so its logits are the eager logits bit for bit — for a new input, after the weights
    moved (the packed copies are refreshed outside of the graph, as after an EMA step), and for the two-sub-batch form;
    HIAST_GRAPH_EVAL=0: nothing is captured; unset (round 6): small batches only."""
    from hiast_amd import functional as HF
    monkeypatch.setenv("HIAST_GRAPH_EVAL", "1")
    cfg, net = _model(782)
    net.eval()
    xs = [torch.from_numpy(synth.normal_f32(940 + i, (4, 3, 64, 128))).cuda() for i in range(5)]

    for parts in (1, 2):
        g = HF.GraphedEval(net, amp, parts=parts)
        for i, x in enumerate(xs):
            if i == 4:          # weights move: replay must see them
                for p in net.parameters():
                    p.data.mul_(1.01)
                torch.autograd.graph.increment_version(list(net.parameters()))
            out = g(x).clone()
            ref = g(x, eager=True)
            torch.cuda.synchronize()
            assert out.dtype == torch.float32 and out.shape == ref.shape
            assert torch.equal(out, ref), (parts, i, float((out - ref).abs().max()))
        e = list(g.entries.values())
        assert len(e) == 1 and e[0]["graph"] is not None and not e[0].get("failed"), "the forward was never captured"
    assert not torch.equal(g(xs[0]).clone(), g(xs[1]))

    monkeypatch.setenv("HIAST_GRAPH_EVAL", "0")
    g = HF.GraphedEval(net, amp)
    for x in xs[:4]:
        g(x)
    assert not g.entries
    # unset + HIAST_GRAPH_EVAL_MAX_BATCH=n: automatic — forwards over at most n images are replayed, larger batches stay eager
    # (off by default: measured slower end to end on this runtime, profiles/r06_ab_graph_eval_small_batch.txt)
    monkeypatch.delenv("HIAST_GRAPH_EVAL")
    assert HF.GraphedEval.AUTO_MAX_BATCH == 0 and not HF.GraphedEval(net, amp).wants_graph(xs[0])
    monkeypatch.setattr(HF.GraphedEval, "AUTO_MAX_BATCH", 4)
    g = HF.GraphedEval(net, amp)
    assert g.enabled is None and g.wants_graph(xs[0]) and not g.wants_graph(torch.empty(8, 3, 64, 128))
    big = torch.cat([xs[0], xs[1]], 0)                      # 8 images
    for _ in range(4):
        g(big)
    assert not g.entries
    outs = [g(xs[0]).clone() for _ in range(4)]
    e = list(g.entries.values())
    assert len(e) == 1 and e[0]["graph"] is not None
    ref = g(xs[0], eager=True)
    assert all(torch.equal(o, ref) for o in outs)


@pytest.mark.parametrize("shape", [(2, 64, 37, 53), (1, 64, 64, 128), (3, 32, 8, 9)])
def test_stem_tail_equals_the_three_passes(shape):
    """K9f hiast_stem_tail (bn1 eval + ReLU + 3x3/2 max pooling + operand format in one pass) against the passes it replaces
    — hiast_bn_act_nhwc_infer, torch's MaxPool2d(3, 2, 1), hiast_split_planes / the bf16 cast — bit for bit, odd sizes
    included; and the whole inference forward with and without it"""
    from hiast_amd import kernels as K
    B, C, H, W = shape
    dev = torch.device("cuda")
    bn = torch.nn.BatchNorm2d(C).to(dev).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(synth.normal_f32(961, (C,), 0.5)) + 1.0)
        bn.bias.copy_(torch.from_numpy(synth.normal_f32(962, (C,), 0.3)))
        bn.running_mean.copy_(torch.from_numpy(synth.normal_f32(963, (C,), 0.4)))
        bn.running_var.copy_(torch.from_numpy(synth.normal_f32(964, (C,), 0.2)).abs() + 0.5)
    x = torch.from_numpy(synth.normal_f32(960, shape, 1.5)).to(dev).contiguous(memory_format=torch.channels_last)
    pool = torch.nn.MaxPool2d(3, 2, 1)
    for xin, PL in ((x, 2), (x.bfloat16(), 1), (x, 1)):
        x2d = xin.permute(0, 2, 3, 1).reshape(B * H * W, C)
        a = K.bn_act_nhwc_infer(x2d, bn, True).view(B, H, W, C).permute(0, 3, 1, 2)
        p = pool(a).contiguous(memory_format=torch.channels_last)
        Ho, Wo = p.shape[2:]
        p2d = p.permute(0, 2, 3, 1).reshape(B * Ho * Wo, C)
        ref = K.split_planes(p2d.float()) if PL == 2 else p2d.to(torch.bfloat16)
        got = K.stem_tail(xin, bn, PL)
        assert got.shape == (B, Ho, Wo, PL * C)
        assert torch.equal(got.view(B * Ho * Wo, PL * C).view(torch.int16), ref.view(torch.int16)), (shape, PL, xin.dtype)


def test_inference_forward_with_and_without_the_fused_stem_tail(monkeypatch):
    cfg, net = _model(783)
    net.eval()
    x = torch.from_numpy(synth.normal_f32(965, (2, 3, 65, 97))).cuda()
    for amp in (False, True):
        outs = []
        for off in ("0", "1"):
            monkeypatch.setenv("HIAST_NO_STEM_TAIL", off)
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                outs.append(net(x, lowres=True)["logits_lowres"].float().clone())
        assert torch.equal(outs[0], outs[1]), amp


@pytest.mark.parametrize("shape", [(2, 64, 37, 53), (1, 64, 64, 128), (2, 16, 5, 4)])
def test_stem_max_pooling_kernels_equal_the_library(shape):
    """K18 (hiast_maxpool3x3s2_nhwc_fwd / _bwd through HF.maxpool): values and input gradient of nn.MaxPool2d(3, 2, 1) on
    channels-last bf16, bit for bit — with ties (coarsely quantised values: the first maximum of a window takes the
    gradient), odd sizes, and a few -inf / NaN entries"""
    from hiast_amd import functional as HF
    B, C, H, W = shape
    dev = torch.device("cuda")
    pool = torch.nn.MaxPool2d(3, 2, 1)
    for case in ("smooth", "ties", "special"):
        v = torch.from_numpy(synth.normal_f32(970, shape, 1.0)).to(dev)
        if case == "ties":
            v = torch.round(v * 2) / 2
        if case == "special":
            v.view(-1)[::97] = float("-inf")
            v.view(-1)[5::389] = float("nan")
        x = v.bfloat16().contiguous(memory_format=torch.channels_last)
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ya = HF.maxpool(xa, pool)
        yb = pool(xb)
        assert ya.dtype == torch.bfloat16 and ya.shape == yb.shape and ya.permute(0, 2, 3, 1).is_contiguous()
        assert torch.equal(ya.detach().view(torch.int16), yb.detach().contiguous(memory_format=torch.channels_last).view(torch.int16)), case
        g = torch.from_numpy(synth.normal_f32(971, tuple(yb.shape), 1.0)).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
        ya.backward(g)
        yb.backward(g)
        ga, gb = xa.grad, xb.grad.contiguous(memory_format=torch.channels_last)
        if case == "special":       # where the window holds a NaN the library's pick is what we copy; compare the finite rest
            ok = ~(torch.isnan(ga.float()) | torch.isnan(gb.float()))
            assert torch.equal(torch.isnan(ga.float()), torch.isnan(gb.float()))
            assert torch.equal(ga.float()[ok], gb.float()[ok]), case
        else:
            assert torch.equal(ga.view(torch.int16), gb.view(torch.int16)), case
