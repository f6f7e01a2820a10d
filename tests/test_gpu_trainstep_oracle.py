"""The ASSEMBLED self-training step against the oracle (not against this package's other paths):

    TRAINER['ConsistencySelfTrainingTrainer'].train_on  (EMA-teacher eval forward on the weak view, student train-mode
    forward on the strong view, fused 4-term region-adaptive loss)  ->  g_loss = sum(mean(loss_i))  ->  backward
    (reference: workflows/trainer/consistency_self_training_trainer.py:92-126, sseg/models/segmentors/
    self_training_segmentor.py:30-53, workflows/trainer/base_trainer.py:127-141)

vs oracle/deeplab_ref.deeplab_v2(train=True) + oracle/losses_ref.st_losses under CPU autograd in fp32 on the same seeded
weights and inputs: the four loss values, EVERY gradient tensor (104 trunk convolutions + 4 x (weight, bias) of the
head; BatchNorm affine is frozen, utils/utils.py:60-65) and the BatchNorm running statistics the train-mode forward
leaves behind.

Modes: O0 = fp32 (library convolutions + own fp32 BN / ASPP / loss kernels): tight bounds.
       O1/bf16, O1/fp16 = the mixed-precision step on the hand-written channels-last kernels: the head and the loss
       must still agree closely; deep in the trunk a 16-bit forward on random-init weights with batch-statistics BN
       cannot reproduce fp32 gradients element by element (two fp32 implementations already differ there), so the
       per-tensor cosines are RECORDED (gpurun_out/r03_trainstep_oracle_<mode>.txt -> profiles/) and bounded loosely.
"""
import os

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W, C, B = 128, 256, 19, 2
WEIGHTS = dict(w_t=1.0, w_k=0.1, w_e=1.0, w_c=0.5)


def _state(seed=9500):
    """seeded weights whose BatchNorm running statistics are those of the data (the teacher normalises with them) and
    whose head is scaled to logits of a few units, so that all four loss terms and their gradients are healthy"""
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import MODEL
    from hiast_amd.utils.default_config import get_default_cfg
    from make_golden import seeded_state_dict
    from oracle import deeplab_ref
    m = MODEL["SelfTrainingSegmentor"](get_default_cfg())
    sd = seeded_state_dict(m.seg_model, seed)
    x = torch.from_numpy(synth.normal_f32(seed + 1, (B, 3, H, W)))
    with torch.no_grad():
        for _ in range(1):          # running statistics := batch statistics of the synthetic data (CPU oracle)
            so = {}
            deeplab_ref.deeplab_v2(x, sd, train=True, stats_out=so)
            for k, v in so.items():
                sd[k] = sd[k] + (v - sd[k]) / 0.1
        s = 4.0 / float(deeplab_ref.deeplab_v2(x, sd)[0].std())
    for i in range(4):
        sd["aspp.conv2d_list.%d.weight" % i] = sd["aspp.conv2d_list.%d.weight" % i] * s
        sd["aspp.conv2d_list.%d.bias" % i] = sd["aspp.conv2d_list.%d.bias" % i] * s
    return {"seg_model." + k: v for k, v in sd.items()}


def _inputs():
    weak = synth.normal_f32(9601, (B, 3, H, W), 1.0)
    strong = (weak * 1.05 + 0.02 + synth.normal_f32(9602, (B, 3, H, W), 0.05)).astype(np.float32)
    plbl = synth.pseudo_labels(9603, B, H, W, C, 0.4)
    return weak, strong, plbl


@pytest.fixture(scope="module")
def oracle_step(tmp_path_factory):
    """one CPU-autograd step of the oracle -> losses, gradients, post-forward running statistics; + the checkpoint file"""
    from oracle import deeplab_ref, losses_ref
    root = str(tmp_path_factory.mktemp("trainstep"))
    sd = _state()
    torch.save(sd, os.path.join(root, "init.pth"))
    weak, strong, plbl = _inputs()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sub = {k[len("seg_model."):]: v for k, v in sd.items()}
    with torch.no_grad():
        zt = deeplab_ref.deeplab_v2(torch.from_numpy(weak), sub, train=False)[0]       # EMA teacher == student at step 1
    params = {k: v.detach().clone().requires_grad_(v.dim() == 4 or "aspp" in k) for k, v in sub.items()}
    so = {}
    zs = deeplab_ref.deeplab_v2(torch.from_numpy(strong), params, train=True, stats_out=so)[0]
    L = losses_ref.st_losses(zs, zt, torch.from_numpy(plbl.astype(np.int64)), (H, W), "ignored", dtype=torch.float32,
                             **WEIGHTS)
    sum(L.values()).backward()
    grads = {"seg_model." + k: p.grad.numpy() for k, p in params.items() if p.grad is not None}
    return {"root": root, "losses": {k: float(v) for k, v in L.items()}, "grads": grads,
            "stats": {"seg_model." + k: v.numpy() for k, v in so.items()}, "zs": zs.detach().numpy()}


def _trainer(root, apex_opt, amp_dtype):
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.default_config import get_default_cfg
    from hiast_amd.workflows.trainer.consistency_self_training_trainer import ConsistencySelfTrainingTrainer

    class StepOnly(ConsistencySelfTrainingTrainer):       # the real trainer minus its disk-backed data readers
        def assert_cfg(self):
            pass

        def build_train_data_reader(self):
            pass

        def build_val_data_reader(self):
            self.v_loader = None

    c = get_default_cfg()
    c.trainer = "ConsistencySelfTrainingTrainer"
    c.model.type = "SelfTrainingSegmentor"
    c.model.predictor.kld_loss.weight = WEIGHTS["w_k"]
    c.model.predictor.ent_loss.weight = WEIGHTS["w_e"]
    c.cst_training.is_enabled = True
    c.cst_training.cst_loss.weight = WEIGHTS["w_c"]
    c.cst_training.cst_loss.region = "ignored"
    c.train.lr, c.train.optimizer, c.train.total_iter = 1e-3, "Adam", 10
    c.train.apex_opt = apex_opt
    c.train.amp_dtype = amp_dtype
    c.train.gpu_num = 1
    c.train.resume_from = os.path.join(root, "init.pth")
    c.work_dir = os.path.join(root, "work_%s_%s" % (apex_opt, amp_dtype))
    c.freeze()
    return StepOnly(c, 0)


def _cos(a, b):
    a, b = a.ravel().astype(np.float64), b.ravel().astype(np.float64)
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


def _device_step(tr):
    from hiast_amd import functional as HF
    weak, strong, plbl = _inputs()
    dev = tr.device
    losses = tr.train_on(torch.from_numpy(weak).to(dev), torch.from_numpy(strong).to(dev), torch.from_numpy(plbl).to(dev))
    g_loss = sum(torch.mean(v) for v in losses.values())
    net = tr.model.module
    scale = 1.0
    if tr.scaler is not None:           # fp16: apex-style loss scaling; halve on overflow like the scaler would
        scale = 2.0 ** 16
    while True:
        tr.g_optimizer.zero_grad(set_to_none=True)
        HF.enable_wgrad_overlap(tr.wgrad_overlap)
        try:
            (g_loss * scale).backward(retain_graph=tr.scaler is not None)
        finally:
            HF.enable_wgrad_overlap(False)
        HF.wgrad_stream_join()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().float().cpu().numpy() / scale for k, p in net.named_parameters() if p.grad is not None}
        if tr.scaler is None or all(np.isfinite(g).all() for g in grads.values()) or scale <= 1.0:
            break
        scale *= 0.5
    stats = {k: v.detach().float().cpu().numpy() for k, v in net.state_dict().items()
             if k.endswith(("running_mean", "running_var"))}
    return {k: float(v) for k, v in losses.items()}, grads, stats, scale


MODES = {"O0": ("O0", "bf16"), "O1_bf16": ("O1", "bf16"), "O1_fp16": ("O1", "fp16")}


@pytest.mark.parametrize("mode", list(MODES))
def test_training_step_vs_cpu_autograd_oracle(oracle_step, mode):
    tr = _trainer(oracle_step["root"], *MODES[mode])
    losses, grads, stats, scale = _device_step(tr)
    fp32 = mode == "O0"
    want = oracle_step["losses"]
    lines = ["training step vs CPU-autograd oracle, mode %s, B=%d 3x%dx%d, loss scale %g" % (mode, B, H, W, scale)]
    for k, v in want.items():
        rel = abs(losses[k] - v) / max(1.0, abs(v))
        lines.append("loss %-22s device %.7f oracle %.7f rel %.2e" % (k, losses[k], v, rel))
        assert rel <= (2e-4 if fp32 else 3e-2), (k, losses[k], v)
    og = oracle_step["grads"]
    assert set(grads) == set(og) and len(og) == 112, (len(grads), len(og))
    rel, cos = {}, {}
    for k in og:
        cos[k] = _cos(grads[k], og[k])
        rel[k] = float(np.abs(grads[k] - og[k]).max() / (np.abs(og[k]).max() + 1e-30))
    order = [k for k in tr.model.module.state_dict() if k in og]
    for k in order:
        lines.append("grad %-52s cos %.6f  max-rel %.2e  |g|max %.3e" % (k[len("seg_model."):], cos[k], rel[k],
                                                                       float(np.abs(og[k]).max())))
    head = [k for k in og if "aspp" in k]
    trunk = [k for k in og if "aspp" not in k]
    lines.append("summary: head cos min %.6f, trunk cos min %.6f mean %.6f; max-rel head %.2e trunk max %.2e median %.2e"
                 % (min(cos[k] for k in head), min(cos[k] for k in trunk), float(np.mean([cos[k] for k in trunk])),
                    max(rel[k] for k in head), max(rel[k] for k in trunk), float(np.median([rel[k] for k in trunk]))))
    # running statistics after the train-mode forward (momentum 0.1, unbiased variance; the student's 104 layers)
    srel = {}
    for k, v in oracle_step["stats"].items():
        srel[k] = float(np.abs(stats[k] - v).max() / (np.abs(v).max() + 1e-30))
    lines.append("running statistics: max rel deviation %.2e (%s)" % (max(srel.values()), max(srel, key=srel.get)))
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "r03_trainstep_oracle_%s.txt" % mode), "w") as f:
            f.write("\n".join(lines) + "\n")
    except OSError:
        pass
    print("\n".join(lines[:6] + lines[-2:]))
    if fp32:
        assert min(cos.values()) >= 0.999, min(cos.items(), key=lambda kv: kv[1])
        assert max(rel[k] for k in head) <= 1e-3, max(((k, rel[k]) for k in head), key=lambda kv: kv[1])
        assert max(rel.values()) <= 1e-2, max(rel.items(), key=lambda kv: kv[1])
        assert max(srel.values()) <= 1e-3, max(srel.items(), key=lambda kv: kv[1])
    else:
        assert min(cos[k] for k in head) >= 0.995, min(((k, cos[k]) for k in head), key=lambda kv: kv[1])
        assert cos["seg_model.backbone.layer4.2.conv3.weight"] >= 0.98
        assert min(cos[k] for k in trunk if "layer4" in k) >= 0.9
        assert max(srel.values()) <= 5e-2, max(srel.items(), key=lambda kv: kv[1])
