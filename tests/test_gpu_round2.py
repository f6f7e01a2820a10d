"""Round-2 regression tests of the host side of the HIP path."""
import os

import numpy as np
import pytest
import torch

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
    adj = conv.__dict__["_hiast_packed_adj"]
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


def test_copy_paste_on_device_matches_reference_golden(golden):
    """K17: the hard-aware CopyPaste composite as one kernel on device-resident frames, bit for bit against the outputs
    of the reference's CopyPaste.run (tests/golden/copy_paste.npz) and against the host path on ragged sizes"""
    from hiast_amd import kernels as K
    from hiast_amd.utils.default_config import get_default_cfg
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import PREPROCESSOR
    g = golden("copy_paste")
    N, H, W, C = [int(v) for v in g["shape"]]
    imgs = synth.images_u8(1100, N, H, W)
    lbls = np.stack([synth.pseudo_labels(1110 + i, 1, H, W, C, 0.3)[0] for i in range(N)])
    names = ["img_%d.png" % i for i in range(N)]
    swc = {c: [names[i] for i in range(N) if (lbls[i] == c).any()] for c in range(C)}

    class DS:
        def get_samples_with_class(self):
            return swc

        def get_file_to_idx(self, f):
            return names.index(f)

        def load_data(self, i):
            return imgs[i].copy(), lbls[i].copy(), names[i]

    c = get_default_cfg()
    c.dataset.source.type = "GTAV"
    cp = PREPROCESSOR["CopyPaste"](c, DS(), g["class_value"].copy())
    np.random.seed(888)
    for i in range(N):
        im, lb, mk = cp.run_device(torch.from_numpy(imgs[i].copy()).cuda(), torch.from_numpy(lbls[i].copy()).cuda())
        assert np.array_equal(im.cpu().numpy(), g["img"][i]) and np.array_equal(lb.cpu().numpy(), g["lbl"][i])
        assert np.array_equal(mk.cpu().numpy(), g["mask"][i])
    # full-size frame (2048 x 1024), random hard set: against the definition
    Hh, Ww = 1024, 2048
    a, b = synth.images_u8(1, 1, Hh, Ww)[0], synth.images_u8(2, 1, Hh, Ww)[0]
    la, lb_ = synth.pseudo_labels(3, 1, Hh, Ww, 19, 0.3)[0], synth.pseudo_labels(4, 1, Hh, Ww, 19, 0.3)[0]
    hard = [0, 3, 7, 18]
    sel = np.isin(lb_, hard)
    da, dl = torch.from_numpy(a.copy()).cuda(), torch.from_numpy(la.copy()).cuda()
    mk = K.copy_paste_u8(da, dl, torch.from_numpy(b).cuda(), torch.from_numpy(lb_).cuda(), hard)
    assert np.array_equal(da.cpu().numpy(), np.where(sel[..., None], b, a))
    assert np.array_equal(dl.cpu().numpy(), np.where(sel, lb_, la)) and np.array_equal(mk.cpu().numpy(), np.where(sel, lb_, 255))
