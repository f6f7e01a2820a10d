"""The two parity size gaps the round-4 review named, closed at BASELINE sizes on the MI355X:

(a) configs[2]: the ASSEMBLED training step at 512x1024 (ResNet-101, apex-O1 fp16, B = 1) against the float64 oracle evaluated
    around the device's own gates — until round 4 the assembled step met the oracle at 128x256 only, so the shape-gated kernels
    (xconv from M >= 4096, the grouped weight gradients from chip fill 0.9, the persistent BatchNorm grids) met each other in the
    bench only (reference: workflows/trainer/consistency_self_training_trainer.py:92-126, base_trainer.py:127-141);
(b) configs[1]: PSEUDO_POLICY['IAS'](cfg).run() over 500 synthetic 1024x512 target images at the reference's batch size 2
    (workflows/pseudo_label_generator.py:181-213): the first three batches replayed through the oracle (bit-equal label maps),
    conservation and artefact checks on all 500, end-to-end images/s printed."""
import json
import os
import time

import numpy as np
import pytest
import torch

import synth
import test_gpu_trainstep_oracle as TS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record(name, lines):
    print("\n".join(lines))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", name), "w") as f:
            f.write("\n".join(lines) + "\n")
    except OSError:
        pass


@pytest.mark.parametrize("batch", [1, 2])
def test_training_step_at_config2_size_given_the_device_gates(tmp_path_factory, monkeypatch, batch):
    """B = 1 and B = 2 (round 6: cross-image batch statistics, B-dependent launch plans), 3x512x1024, ResNet-101, O1 fp16: losses
    and all 112 gradient tensors against the float64 oracle around the device's gates, same per-tensor bounds as at 128x256
    (test_gpu_trainstep_oracle.GATED_BOUNDS); beside it the FREE-RUNNING check (the oracle on its own gates, float64): the
    student's low-resolution logits and the four losses — what the gated comparison cannot see is a forward that is wrong in a
    way its own gates follow"""
    for k, v in (("H", 512), ("W", 1024), ("B", batch)):
        monkeypatch.setattr(TS, k, v)
    depth, mode = "r101", "O1_fp16"
    TS._patch_depth(monkeypatch, depth)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    t0 = time.time()
    root = str(tmp_path_factory.mktemp("trainstep_fullsize"))
    sd = TS._state(depth)                               # (two float32 oracle forwards at this size)
    torch.save(sd, os.path.join(root, "init.pth"))
    tr = TS._trainer(root, *TS.MODES[mode])
    masks, pool = TS._capture_gates(monkeypatch, tr.model.module)
    seen = {}
    loss_fn = tr.model.module.compute_loss_lowres

    def spy(z, *a, **k):                                # the student's low-resolution logits as the loss kernel receives them
        seen["z"] = z.detach().float().cpu().numpy()
        return loss_fn(z, *a, **k)
    monkeypatch.setattr(tr.model.module, "compute_loss_lowres", spy)
    losses, grads, _stats, scale = TS._device_step(tr)
    assert len(masks) == 1 + 3 * sum(TS.DEPTHS[depth]) and "codes" in pool
    t1 = time.time()
    want, og = TS._oracle_step_given_gates(sd, masks, pool["codes"])
    t2 = time.time()
    lines = ["training step vs float64 oracle AROUND THE DEVICE'S GATES at BASELINE configs[2] size: mode %s, trunk %s, B=%d "
             "3x%dx%d, loss scale %g (state + device step %.0f s, float64 oracle %.0f s)"
             % (mode, depth, TS.B, TS.H, TS.W, scale, t1 - t0, t2 - t1)]
    for k, v in want.items():
        lines.append("loss %-22s device %.7f gated oracle %.7f rel %.2e" % (k, losses[k], v, abs(losses[k] - v) / max(1.0, abs(v))))
    order = [k for k in tr.model.module.state_dict() if k in og]
    assert set(order) == set(grads) and len(order) == 112
    cos, rel, l2 = {}, {}, {}
    for k in order:
        cos[k] = TS._cos(grads[k], og[k])
        rel[k] = float(np.abs(grads[k] - og[k]).max() / (np.abs(og[k]).max() + 1e-30))
        l2[k] = float(np.linalg.norm(grads[k] - og[k]) / (np.linalg.norm(og[k]) + 1e-30))
        lines.append("grad %-52s cos %.8f  max-rel %.2e  rel-L2 %.2e" % (k[len("seg_model."):], cos[k], rel[k], l2[k]))
    lines.append("summary: cos min %.8f (%s) mean %.8f; max-rel max %.2e median %.2e; rel-L2 max %.2e"
                 % (min(cos.values()), min(cos, key=cos.get)[len("seg_model."):], float(np.mean(list(cos.values()))),
                    max(rel.values()), float(np.median(list(rel.values()))), max(l2.values())))
    # free-running: the oracle's own forward (its own ReLU gates and pooling windows) in float64 — teacher in eval mode, student
    # in train mode — and the reference's four losses on it
    from oracle import deeplab_ref, losses_ref
    weak, strong, plbl = TS._inputs()
    sub = {k[len("seg_model."):]: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    with torch.no_grad():
        zt = deeplab_ref.deeplab_v2(torch.from_numpy(weak).double(), sub, train=False)[0]
        zs = deeplab_ref.deeplab_v2(torch.from_numpy(strong).double(), sub, train=True, stats_out={})[0]
        Lf = losses_ref.st_losses(zs, zt, torch.from_numpy(plbl.astype(np.int64)), (TS.H, TS.W), "ignored", dtype=torch.float64,
                                  **TS.WEIGHTS)
    t3 = time.time()
    zerr = float(np.abs(seen["z"] - zs.numpy()).max() / np.abs(zs.numpy()).max())
    zrms = float(np.sqrt(np.mean((seen["z"] - zs.numpy()) ** 2)) / np.sqrt(np.mean(zs.numpy() ** 2)))
    lines.append("free-running float64 oracle (own gates; %.0f s): student low-res logits max-err %.3e of max|z|, rms-err %.3e of rms"
                 % (t3 - t2, zerr, zrms))
    for k, v in Lf.items():
        lines.append("free-running loss %-22s device %.7f oracle %.7f rel %.2e" % (k, losses[k], float(v),
                                                                                      abs(losses[k] - float(v)) / max(1.0, abs(float(v)))))
    _record("r06_trainstep_gated_oracle_r101_O1_fp16_512x1024_B%d.txt" % batch, lines)
    # fp16 storage of 100 activation tensors between the input and the logits, each renormalised by batch statistics: measured
    # 1.7e-2 (B = 2) / 2.0e-2 (B = 1) of max|z|, rms 1.8e-2 of the rms, losses within 5e-5 (profiles/r06_trainstep_gated_oracle_*);
    # an O(1) error in one layer (wrong statistic, dropped tap, wrong scale) moves the logits by tens of percent
    assert zerr <= 6e-2 and zrms <= 4e-2, (zerr, zrms)
    for k, v in Lf.items():
        assert abs(losses[k] - float(v)) <= 3e-2 * max(1.0, abs(float(v))), ("free-running", k, losses[k], float(v))
    for k, v in want.items():
        assert abs(losses[k] - v) <= 3e-2 * max(1.0, abs(v)), (k, losses[k], v)
    lo_cos, hi_rel = TS.GATED_BOUNDS[depth][mode]
    worst = min(cos.items(), key=lambda kv: kv[1])
    assert worst[1] >= lo_cos, worst
    worst = max(rel.items(), key=lambda kv: kv[1])
    assert worst[1] <= hi_rel, worst


def test_config1_generator_500_images_1024x512(tmp_path_factory):
    from PIL import Image
    from oracle import cref, ias_ref
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import MODEL, PSEUDO_POLICY
    from hiast_amd.sseg.datasets import utils as du
    from hiast_amd.tools import synth_data
    from hiast_amd.utils import comm
    from make_golden import seeded_state_dict
    # NOTE: step (1) below replays the oracle from the DEVICE's logits: it pins everything behind the forward (upsample, softmax,
    # fp16 lists, quantile, EMA, selection, artefacts) bit for bit.  Parity of the FORWARD at this size is test_gpu_fullsize.py's
    # (device logits against oracle/deeplab_ref.py at 512x1024 and 1024x2048: <= 1e-3 of max, argmax diff 0).
    N, H, W, C, BS = 500, 512, 1024, 19, 2
    root = str(tmp_path_factory.mktemp("config1"))
    workers = max(2, min(14, comm.usable_cpus() - 2))
    t0 = time.time()
    cfg = synth_data.synthetic_cfg(root, n_train=N, n_val=1, h=H, w=W, procs=workers)
    t_write = time.time() - t0
    # a checkpoint in the state of a trained one (tests/test_gpu_e2e.py): running statistics of the data, head calibrated to
    # logits of std 3 so that the confident set is neither empty nor everything
    m = MODEL["SelfTrainingSegmentor"](cfg)
    sd = {"seg_model." + k: v for k, v in seeded_state_dict(m.seg_model, 777).items()}
    m.load_state_dict(sd)
    m = m.cuda().eval()
    ds = np.stack([synth_data.make_sample(5 + i, H, W)[0].astype(np.float32).transpose(2, 0, 1) for i in range(2)]) / 255.0
    xs = torch.from_numpy((ds - 0.45) / 0.225).cuda()
    synth_data.calibrate_bn(m, xs)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        scale = 3.0 / float(m(xs[:1], lowres=True)["logits_lowres"].std())
    for i in range(4):
        sd["seg_model.aspp.conv2d_list.%d.weight" % i] = sd["seg_model.aspp.conv2d_list.%d.weight" % i] * scale
        sd["seg_model.aspp.conv2d_list.%d.bias" % i] = sd["seg_model.aspp.conv2d_list.%d.bias" % i] * scale
    del m
    ck = os.path.join(root, "warmup.pth")
    torch.save(sd, ck)
    cfg.pseudo_policy.resume_from = ck
    cfg.pseudo_policy.batch_size = BS
    cfg.dataset.num_workers = workers
    gen = PSEUDO_POLICY["IAS"](cfg)
    gen.engine.pass1(torch.zeros((BS, H, W, 3), dtype=torch.uint8))       # kernel load, not counted
    gen.engine.pass2(None)
    torch.cuda.synchronize()
    t0 = time.time()
    gen.run()
    torch.cuda.synchronize()
    dt = time.time() - t0
    pdir = cfg.pseudo_policy.save_dir
    files = sorted(os.listdir(pdir))
    assert len(files) == N
    thr = np.load(os.path.join(pdir, "..", "class_threshold.npy"))
    stats = np.load(os.path.join(pdir, "..", "statics_class.npy"))
    assert thr.shape == (C,) and thr.dtype == np.float64 and bool(((thr > 0) & (thr < 1)).all())

    # (1) the first three batches through the ORACLE (same device logits, then stage A in C + the reference's list / np.quantile
    # IAS step on the host): label maps bit-equal
    st = ias_ref.IASState(C, cfg.pseudo_policy.ias.alpha, cfg.pseudo_policy.ias.beta, cfg.pseudo_policy.ias.gamma, 0.99)
    model = gen.engine.model
    replayed = 0
    for bi, data in enumerate(gen.t_loader):
        if bi == 3:
            break
        imgs = data["images"]
        if imgs.dtype == torch.uint8:
            imgs = torch.stack([du._img_to_tensor(i.numpy(), du.MEAN, du.STD) for i in imgs])
        with torch.no_grad():
            z = model(imgs.cuda(), lowres=True)["logits_lowres"].float().cpu().numpy()
        mp, am = cref.plabel_stage_a(z, H, W)
        plbl = st.step(mp, am.astype(np.int64), data["image_paths"])
        for b, p in enumerate(data["image_paths"]):
            name = os.path.splitext(os.path.basename(p))[0] + "_pseudo_label.png"
            got = np.array(Image.open(os.path.join(pdir, name)))
            assert got.shape == (H, W) and np.array_equal(got, plbl[b]), name
            replayed += 1
    assert replayed == 3 * BS

    # (2) conservation over all 500 label maps: values in [0, C) + 255, per-class counts of the kept pixels = statics_class.npy
    counts = np.zeros(C, dtype=np.int64)
    kept = 0
    for f in files:
        a = np.array(Image.open(os.path.join(pdir, f)))
        assert a.shape == (H, W) and a.dtype == np.uint8
        bc = np.bincount(a.ravel(), minlength=256)
        assert bc[C:255].sum() == 0, f
        counts += bc[:C]
        kept += int(bc[:C].sum())
    assert np.array_equal(counts, stats.astype(np.int64)), (counts, stats)
    frac = kept / float(N * H * W)
    assert 0.02 < frac < 0.98, frac
    # (3) artefacts
    recs = json.load(open(os.path.join(pdir, "..", "sample_class_stats.json")))
    by_class = json.load(open(os.path.join(pdir, "..", "samples_with_class.json")))
    assert len(recs) == N and len(by_class) == C
    means = np.load(os.path.join(pdir, "..", "class_mean_probabilities.npy"))
    assert means.shape == (C,) and bool(np.isfinite(means).all())
    _record("r06_config1_generator_500.txt", [
        "BASELINE configs[1] as a -m gpu test: PSEUDO_POLICY['IAS'](cfg).run() over %d synthetic %dx%d PNGs, batch size %d (the "
        "reference's), %d DataLoader workers" % (N, W, H, BS, workers),
        "run(): %.2f s = %.1f images/s end to end (PNG decode -> uint8 H2D -> normalise -> fp32-class forward -> pass 1 -> host "
        "thresholds -> pass 2 -> D2H -> PNG files written); dataset written in %.1f s" % (dt, N / dt, t_write),
        "first %d images replayed through the oracle: label maps bit-equal; all %d maps: values in [0,%d) + 255, per-class kept "
        "pixel counts == statics_class.npy, kept fraction %.3f; thresholds %s" % (replayed, N, C, frac, np.round(thr, 4).tolist())])
