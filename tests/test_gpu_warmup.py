"""Adversarial warm-up / source-only stage on the MI355X: the discriminator-input kernel (K15) against
the oracle, MODEL['AdversarialWarmupSegmentor'] against the fixtures produced by the reference itself
(tests/golden/warmup.npz), and the two trainers end to end on a synthetic dataset."""
import os

import numpy as np
import pytest
import torch

import synth
from oracle import warmup_ref
from test_oracle_golden import WARMUP_CASES, warmup_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from hiast_amd import kernels
    return kernels


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


DIN_SHAPES = [(2, 19, 8, 16, 64, 128), (1, 19, 9, 17, 65, 129), (1, 9, 6, 6, 48, 48), (1, 2, 3, 4, 100, 301),
              (1, 16, 5, 7, 5, 7), (2, 19, 64, 128, 512, 1024)]


@pytest.mark.parametrize("entropy", [False, True])
@pytest.mark.parametrize("shape", DIN_SHAPES)
def test_dinput_fwd_bwd_vs_oracle(K, shape, entropy):
    """fp32 kernel vs the oracle (fp32 bilinear coordinates like the reference, float64 softmax / log2): map within
    5e-6 absolute (values are in [0,1]; interpolated fp32 logits of magnitude ~10 carry ~1e-6 of rounding, which
    the softmax passes on); gradient within 1e-4 of its scale (adjoint sums of up to ~16x16 terms in fp32)"""
    B, C, h, w, H, W = shape
    z = synth.logits_lr(31, B, C, h, w, 3.0)
    g = synth.normal_f32(32, (B, C, H, W))
    got = K.dinput_fwd(dev(z), H, W, entropy)
    zt = torch.from_numpy(z).requires_grad_(True)
    want = warmup_ref.discriminator_input(zt, (H, W), entropy, interp_dtype=torch.float32)
    assert np.abs(got.cpu().numpy() - want.detach().numpy()).max() <= 5e-6
    want.backward(torch.from_numpy(g).double())
    d = K.dinput_bwd(dev(z), dev(g), entropy).cpu().numpy()
    ref = zt.grad.numpy()
    assert np.abs(d - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-3)


def test_dinput_underflowed_probability(K):
    """a class whose probability underflows to 0: the self-information map and its gradient stay finite (the
    reference's + 1e-30 inside the log)"""
    B, C, h, w, H, W = 1, 19, 4, 6, 16, 24
    z = synth.logits_lr(33, B, C, h, w, 2.0)
    z[:, 5] = -200.0
    out = K.dinput_fwd(dev(z), H, W, True)
    assert torch.isfinite(out).all() and float(out[:, 5].abs().max()) == 0.0
    d = K.dinput_bwd(dev(z), torch.ones_like(out), True)
    assert torch.isfinite(d).all()


def test_dinput_autograd_and_autocast(K):
    from hiast_amd import functional as HF
    B, C, h, w, H, W = 2, 19, 8, 16, 64, 128
    z = dev(synth.logits_lr(34, B, C, h, w, 3.0)).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        x = HF.discriminator_input(z.bfloat16(), (H, W), False)
        assert x.dtype == torch.float32
        (x * x).sum().backward()
    zt = z.detach().bfloat16().float().cpu().requires_grad_(True)
    xr = warmup_ref.discriminator_input(zt, (H, W), False)
    (xr * xr).sum().backward()
    # the gradient reaches the fp32 leaf through a bf16 tensor: one bf16 rounding (2^-8 relative)
    assert np.allclose(z.grad.cpu().numpy(), zt.grad.numpy(), rtol=1e-2, atol=1e-3)
    assert float(x.sum(1).sub(1).abs().max()) < 1e-5          # a softmax


class _StubSeg(torch.nn.Module):
    def __init__(self, outs):
        super().__init__()
        self.outs, self.i = list(outs), 0

    def forward(self, x, need_feat=True):
        o = self.outs[self.i % len(self.outs)]
        self.i += 1
        return o, None


def _warmup_cfg(cs, C):
    from hiast_amd.utils.default_config import get_default_cfg
    c = get_default_cfg()
    c.dataset.num_classes = C
    c.model.type = "AdversarialWarmupSegmentor"
    c.model.discriminator.is_enabled = True
    c.model.discriminator.is_entropy_input = cs["entropy_in"]
    c.model.discriminator.D_loss.type = cs["d_loss"]
    c.model.discriminator.D_loss.weight = 1.0
    c.model.discriminator.D_loss.adv_weight = 0.05
    c.model.predictor.seg_loss.source_weight = 1.0
    c.model.predictor.ent_loss.weight = cs["ent_w"]
    return c


@pytest.mark.parametrize("tag", WARMUP_CASES)
def test_warmup_segmentor_vs_reference(K, golden, tag):
    """AdversarialWarmupSegmentor downstream of the segmentation net against the reference's own outputs: the four
    loss values (rel 2e-5; the discriminator's convolutions are MIOpen fp32), the generator-step gradient w.r.t.
    both low-res logit maps and the discriminator-step gradients (1e-3 of each tensor's scale)."""
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import MODEL
    g = golden("warmup")
    cs, (B, C, h, w, H, W), zs, zt, lbl, d_sd = warmup_case(g, tag)
    m = MODEL["AdversarialWarmupSegmentor"](_warmup_cfg(cs, C))
    zs = dev(zs).requires_grad_(True)
    zt = dev(zt).requires_grad_(True)
    m.seg_model = _StubSeg([zs, zt])
    m.D.load_state_dict(d_sd)
    m = m.cuda().train()
    img = torch.zeros(B, 3, H, W, device="cuda")
    losses = m(img, img, dev(lbl))
    names = ["source_seg_loss", "adv_loss", "D_loss", "target_ent_loss"]
    want = g["vals_" + tag]
    assert set(losses) == {n for n, v in zip(names, want) if not np.isnan(v)}
    for n, v in zip(names, want):
        if not np.isnan(v):
            assert abs(losses[n].item() - v) <= 2e-5 * abs(v) + 1e-7, (n, losses[n].item(), v)
    # generator step (base_trainer.py:129-133)
    sum(torch.mean(v) for k, v in losses.items() if "D_" not in k).backward()

    def close(a, b, tol=1e-3):
        return np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-12)

    assert close(zs.grad.cpu().numpy(), g["gs_" + tag])
    assert close(zt.grad.cpu().numpy(), g["gt_" + tag])
    assert all(p.grad is None for p in m.D.parameters())      # detached weights in the adversarial pass
    # discriminator step (:136-141)
    zs.grad = zt.grad = None
    losses["D_loss"].backward()
    assert zs.grad is None and zt.grad is None
    assert close(m.D.conv1.weight.grad.cpu().numpy(), g["gd_conv1_w_" + tag])
    assert close(m.D.classifier.weight.grad.cpu().numpy(), g["gd_cls_w_" + tag])
    bias = np.concatenate([getattr(m.D, n).bias.grad.cpu().numpy().ravel() for n in
                           ("conv1", "conv2", "conv3", "conv4", "classifier")])
    assert close(bias, g["gd_bias_" + tag])
    wsum = np.array([getattr(m.D, n).weight.grad.double().sum().item() for n in
                     ("conv1", "conv2", "conv3", "conv4", "classifier")])
    assert np.allclose(wsum, g["gd_wsum_" + tag], rtol=2e-3, atol=1e-6)
    dmap = K.dinput_fwd(zt.detach(), H, W, cs["entropy_in"]).cpu().numpy()[:, ::6]
    assert np.abs(dmap - g["dmap_" + tag]).max() <= 5e-6


def test_warmup_losses_registry(K):
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import LOSS
    a = dev(synth.normal_f32(41, (2, 1, 16, 32)))
    one = torch.ones_like(a)
    assert abs(LOSS["MSE"](a, one).item() - float(((a - 1) ** 2).mean())) < 1e-6
    want = torch.nn.functional.binary_cross_entropy_with_logits(a.cpu().double(), one.cpu().double()).item()
    assert abs(LOSS["BCEWithLogits"](a, one).item() - want) < 1e-6
    x = dev(synth.normal_f32(42, (2, 19, 8, 8)))
    y = dev(synth.normal_f32(43, (2, 19, 8, 8)))
    want = torch.nn.KLDivLoss()(torch.log_softmax(x.cpu().double(), 1), torch.softmax(y.cpu().double(), 1)).item()
    assert abs(LOSS["KLDIV"](x, y).item() - want) < 1e-6
    with pytest.raises(RuntimeError):
        LOSS["MSE"](a.cpu(), one.cpu())


def _registry_names():
    from make_golden import LOSS_REGISTRY_CASES
    return [c[0] for c in LOSS_REGISTRY_CASES]


@pytest.mark.parametrize("name", _registry_names())
def test_loss_registry_full_argument_surface_vs_reference(K, golden, name):
    """LOSS['CE'|'SoftCE'|'MSE'|'KLDIV'] with per-class weights, refer_labels + region, SoftCE's plain mean and another
    ignore_index — the part of the registry signature (sseg/models/modules/losses.py:10-89) no HIAST config uses — against
    the reference's own outputs (tests/golden/loss_registry.npz): values 2e-5 relative, gradients 1e-4 of their maximum
    (the fused kernel's class where it serves the case); SoftCE scales its target by the weights in place, as the reference"""
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import LOSS
    from test_oracle_golden import _loss_registry_case
    g = golden("loss_registry")
    kind, z, lbl, w, ign, refer, region = _loss_registry_case(name)
    zt = dev(z).requires_grad_(True)
    lt = dev(lbl)
    soft_before = lt.clone() if kind == "SoftCE" else None
    val = LOSS[kind](zt, lt, weights=None if w is None else dev(w), ignore_index=ign,
                     refer_labels=None if refer is None else dev(refer), region=region)
    val.backward()
    want = float(g["val_" + name])
    assert abs(float(val) - want) <= 2e-5 * max(1.0, abs(want)), (float(val), want)
    gw = g["grad_" + name]
    assert np.abs(zt.grad.cpu().numpy() - gw).max() <= 1e-4 * np.abs(gw).max() + 1e-9
    if kind == "SoftCE" and w is not None:
        assert torch.allclose(lt, soft_before * dev(w).view(1, -1, 1, 1))


# ------------------------------------------------------------------------------------------ trainers end to end
H, W, C = 128, 256, 19


@pytest.fixture(scope="module")
def world(tmp_path_factory):
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import MODEL
    from hiast_amd.tools import synth_data
    from make_golden import seeded_state_dict
    root = str(tmp_path_factory.mktemp("warmup"))
    cfg = synth_data.synthetic_cfg(root, n_train=6, n_val=4, h=H, w=W)
    # the 2-3 iteration plumbing tests below use the 16-bit type WITHOUT loss scaling: under fp16 (the default, apex O1) the
    # first ~10 iterations of a run are skipped while the dynamic scale comes down from 2^16, as under apex — covered by
    # tests/test_gpu_fp16.py::test_fp16_training_step_runs_the_trunk_on_the_own_kernels
    cfg.train.amp_dtype = "bf16"
    cfg.dataset.source.type = "Cityscapes"          # the labelled synthetic split doubles as the source domain
    cfg.dataset.source.json_path = cfg.dataset.target.json_path
    cfg.dataset.source.image_dir = cfg.dataset.target.image_dir
    cfg.dataset.source.aug_type = ["PRS-%d-%d" % (H, W)]
    cfg.dataset.target.aug_type = ["PRS-%d-%d" % (H, W)]
    m = MODEL["SourceOnlySegmentor"](cfg)
    sd = {"seg_model." + k: v for k, v in seeded_state_dict(m.seg_model, 778).items()}
    m.load_state_dict(sd)           # running statistics of the data (see synth_data.calibrate_bn: fp16 range)
    m = m.cuda()
    ds = np.stack([synth_data.make_sample(5 + i, H, W)[0].astype(np.float32).transpose(2, 0, 1) for i in range(2)]) / 255.0
    synth_data.calibrate_bn(m, torch.from_numpy((ds - 0.45) / 0.225).cuda())
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    del m
    ck = os.path.join(root, "imagenet_like.pth")
    torch.save(sd, ck)
    cfg.train.resume_from = ck
    cfg.train.gpu_num = 1
    cfg.train.batch_size = 2
    cfg.train.iter_report = 1
    return cfg, sd, root


def test_source_only_trainer(world):
    from hiast_amd.utils.registry.registries import TRAINER
    cfg, sd, root = world
    c = cfg.clone()
    c.trainer = "SourceOnlyTrainer"
    c.model.type = "SourceOnlySegmentor"
    c.train.total_iter = 3
    c.train.iter_val = 3
    c.train.lr = 1e-4
    c.work_dir = os.path.join(root, "work_so")
    c.freeze()
    tr = TRAINER[c.trainer](c, 0)
    first = tr.train()
    assert set(first) == {"seg_loss"} and torch.isfinite(first["seg_loss"])
    p0 = next(tr.model.module.seg_model.aspp.parameters()).detach().clone()
    tr.run()
    assert not torch.equal(p0, next(tr.model.module.seg_model.aspp.parameters()).detach())
    saved = torch.load(os.path.join(c.work_dir, "checkpoints", "model_last.pth"), map_location="cpu")
    assert list(saved.keys()) == list(sd.keys())


@pytest.mark.parametrize("d_loss,entropy_in", [("MSE", False), ("BCEWithLogits", True)])
def test_adversarial_warmup_trainer(world, d_loss, entropy_in):
    from hiast_amd.utils.registry.registries import TRAINER
    cfg, sd, root = world
    c = cfg.clone()
    c.trainer = "AdversarialWarmupTrainer"
    c.model.type = "AdversarialWarmupSegmentor"
    c.model.discriminator.is_enabled = True
    c.model.discriminator.is_entropy_input = entropy_in
    c.model.discriminator.D_loss.type = d_loss
    c.model.predictor.ent_loss.weight = 3.0
    c.train.total_iter = 2
    c.train.iter_val = 2
    c.work_dir = os.path.join(root, "work_adv_" + d_loss)
    c.freeze()
    tr = TRAINER[c.trainer](c, 0)
    assert tr.d_optimizer is not None and len(tr.schedulers) == 2
    net = tr.model.module
    d0 = net.D.conv1.weight.detach().clone()
    s0 = next(net.seg_model.aspp.parameters()).detach().clone()
    losses = tr.train()
    assert set(losses) == {"source_seg_loss", "adv_loss", "D_loss", "target_ent_loss"}
    assert all(torch.isfinite(v) for v in losses.values()), losses
    tr.run()
    assert not torch.equal(d0, net.D.conv1.weight.detach()), "discriminator did not move"
    assert not torch.equal(s0, next(net.seg_model.aspp.parameters()).detach()), "segmentation net did not move"
    saved = torch.load(os.path.join(c.work_dir, "checkpoints", "model_last.pth"), map_location="cpu")
    assert {"D.conv1.weight", "D.classifier.bias", "seg_model.aspp.conv2d_list.0.weight"} <= set(saved.keys())
    # the checkpoint starts the next stage: SelfTrainingSegmentor loads the seg_model.* part (utils.py:76-84)
    from hiast_amd.utils import utils
    c2 = cfg.clone()
    c2.model.type = "SelfTrainingSegmentor"
    m2 = utils.load_model(c2, resume_from=os.path.join(c.work_dir, "checkpoints", "model_last.pth"))
    assert torch.equal(m2.state_dict()["seg_model.aspp.conv2d_list.0.weight"],
                       saved["seg_model.aspp.conv2d_list.0.weight"])


def test_adversarial_trunk_gradients_do_not_race(world):
    """seg_model runs twice per adversarial step, so every trunk weight gets two gradients that autograd sums on the
    main stream: weight gradients must not run on the side stream there (the trainer switches the overlap off).
    The generator-step gradients equal those of a run with the side stream disabled altogether."""
    from hiast_amd.utils.registry.registries import TRAINER
    from hiast_amd import functional as HF
    cfg, sd, root = world
    grads = []
    for tag, env in (("default", None), ("nostream", "1")):
        c = cfg.clone()
        c.trainer = "AdversarialWarmupTrainer"
        c.model.type = "AdversarialWarmupSegmentor"
        c.model.discriminator.is_enabled = True
        c.train.total_iter = 1
        c.train.iter_val = 100
        c.work_dir = os.path.join(root, "work_adv_race_" + tag)
        c.freeze()
        if env is not None:
            os.environ["HIAST_NO_WGRAD_STREAM"] = env
        try:
            tr = TRAINER[c.trainer](c, 0)
            assert tr.wgrad_overlap is False
            s_img = torch.from_numpy(synth.normal_f32(61, (2, 3, H, W))).cuda()
            t_img = torch.from_numpy(synth.normal_f32(62, (2, 3, H, W))).cuda()
            s_lbl = torch.from_numpy(synth.pseudo_labels(63, 2, H, W, C, 0.1)).cuda()
            losses = tr.train_on(s_img, s_lbl, t_img)
            g_loss = sum(torch.mean(v) for k, v in losses.items() if "D_" not in k)
            tr.g_optimizer.zero_grad(set_to_none=True)
            HF.enable_wgrad_overlap(tr.wgrad_overlap)
            try:
                g_loss.backward()
            finally:
                HF.enable_wgrad_overlap(False)
            HF.wgrad_stream_join()
            torch.cuda.synchronize()
            net = tr.model.module.seg_model.backbone
            grads.append({k: p.grad.detach().float().cpu() for k, p in net.named_parameters()
                          if p.grad is not None and "layer3" in k and "conv" in k})
        finally:
            os.environ.pop("HIAST_NO_WGRAD_STREAM", None)
    assert len(grads[0]) >= 60
    # the two runs differ only by the library's atomically accumulated discriminator / strided-conv gradients (a few
    # 1e-3 of a tensor's largest element by the time they reach layer3); a race leaves stale blocks in dW instead
    for k in grads[0]:
        a, b = grads[0][k].double(), grads[1][k].double()
        assert float((a - b).abs().max()) <= 3e-2 * float(b.abs().max()) + 1e-12, k
        assert float((a * b).sum() / (a.norm() * b.norm())) >= 0.9995, k
