"""SURVEY §8 f-3 on the MI355X: the 'CT' / 'NT' / 'CBST' pseudo-label policies and the Validator's multi-scale + flip
test-time augmentation through the HIP kernels, against the oracle (bit-exact: same HIAST-A arithmetic) and against the
fixtures the reference itself produced (tests/golden/policies.npz, tta.npz; torch's own softmax differs from HIAST-A by
a few ulp, so a handful of pixels may land on the other side of a threshold / in another fp16 bin)."""
import json
import os

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------ K3b strided histogram
@pytest.mark.parametrize("shape,C,interval", [((2, 64, 128), 19, 4), ((8, 512, 1024), 19, 4), ((1, 37, 53), 9, 3),
                                             ((3, 40, 60), 19, 1), ((1, 5, 7), 2, 5)])
def test_strided_hist_equals_list_formulation(shape, C, interval):
    """tmp = probs[lbls == c].astype(f16); list_c.extend(tmp[0::interval]) (pseudo_label_generator.py:153-158) as an
    integer histogram, incl. a rank offset (sharded generation) and the per-class totals"""
    from hiast_amd import kernels as K
    from hiast_amd._lib import NBINS
    g = synth.rng(3100 + shape[1])
    mp = (1.0 / C + (1 - 1.0 / C) * np.sqrt(g.random(shape, dtype=np.float32))).astype(np.float32)
    am = g.integers(0, C, size=shape).astype(np.uint8)
    am[..., : shape[2] // 3] = 0                       # long runs of one class (whole waves with equal keys)
    if C > 3:
        am[am == C - 2] = C - 3                        # and a class that never occurs
    for off in (None, g.integers(0, 1000, size=C)):
        want = np.zeros((C, NBINS), np.int64)
        for c in range(C):
            tmp = mp[am == c].astype(np.float16)
            first = 0 if off is None else (-int(off[c])) % interval
            np.add.at(want[c], tmp[first::interval].view(np.uint16).astype(np.int64), 1)
        offd = None if off is None else torch.from_numpy(off.astype(np.int64)).cuda()
        hist, tot = K.plabel_strided_hist(torch.from_numpy(mp).cuda(), torch.from_numpy(am).cuda(), C, interval,
                                          rank_offset=offd, want_totals=True)
        assert np.array_equal(hist.cpu().numpy().astype(np.int64), want)
        assert np.array_equal(tot.cpu().numpy(), np.bincount(am.ravel(), minlength=C)[:C])
    # accumulation into an existing histogram (the policy pools the whole target set)
    h2 = K.plabel_strided_hist(torch.from_numpy(mp).cuda(), torch.from_numpy(am).cuda(), C, interval, hist=hist.clone(),
                               rank_offset=offd)
    assert np.array_equal(h2.cpu().numpy().astype(np.int64), 2 * want)


# ------------------------------------------------------------------------------------------ CT / NT / CBST generators
class _Items(torch.utils.data.Dataset):
    """dataset stub: item i is fixture image pos[i]; the image tensor only carries that index"""

    def __init__(self, pos, H, W):
        self.pos, self.H, self.W = pos, H, W
        self.device_transform = False

    def __len__(self):
        return len(self.pos)

    def __getitem__(self, i):
        k = int(self.pos[i])
        return {"images": torch.full((3, self.H, self.W), float(k)),
                "image_paths": "data/cityscapes/leftImg8bit/train/x/img_%03d_leftImg8bit.png" % k}


class _HeadOnly(torch.nn.Module):
    """model stub: the low-res head output of fixture image k"""

    def __init__(self, z_all, H, W):
        super().__init__()
        self.z_all, self.H, self.W = z_all, H, W

    def forward(self, imgs, lowres=True):
        idx = imgs[:, 0, 0, 0].long()
        return {"logits_lowres": self.z_all[idx].contiguous(), "size": (self.H, self.W)}


@pytest.mark.parametrize("policy", ["CT", "NT", "CBST"])
def test_constant_policies_on_hip(tmp_path, golden, policy, monkeypatch):
    from PIL import Image
    from oracle import cref, ias_ref
    from make_golden import POLICY_SHAPE, policy_inputs
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import PSEUDO_POLICY
    from hiast_amd.utils.default_config import get_default_cfg
    from hiast_amd.workflows.pseudo_label_generator import HipPlabelEngine, ShardedBatchSampler
    g = golden("policies")
    nb, B, C, h, w, H, W = POLICY_SHAPE
    zs = policy_inputs()
    z_all = torch.from_numpy(np.concatenate(zs)).cuda()
    c = get_default_cfg()
    c.pseudo_policy.type = policy
    c.pseudo_policy.batch_size = B
    c.pseudo_policy.ct.threshold = 0.9
    c.pseudo_policy.cbst.p, c.pseudo_policy.cbst.sample_interval = 0.2, 4
    c.pseudo_policy.save_dir = str(tmp_path / "pseudo" / "pseudo_labels")
    c.dataset.num_workers = 0
    # the generator walks a seeded permutation: lay the fixture images out so that it meets them in fixture order
    order = ShardedBatchSampler(nb * B, B, 0, 1, True, c.train.random_seed).order
    pos = np.empty(nb * B, np.int64)
    pos[np.asarray(order)] = np.arange(nb * B)
    monkeypatch.setenv("HIAST_CBST_QUANTILE", "float16")       # numpy-2 evaluation, as recorded in the fixture
    dev = torch.device("cuda", 0)
    gen = PSEUDO_POLICY[policy](c, engine=HipPlabelEngine(_HeadOnly(z_all, H, W), dev, C), dataset=_Items(pos, H, W))
    gen.run()
    root = str(tmp_path / "pseudo")
    got = np.stack([np.array(Image.open(os.path.join(c.pseudo_policy.save_dir, "img_%03d_leftImg8bit_pseudo_label.png" % k)))
                    for k in range(nb * B)])
    # (1) oracle replay in HIAST-A arithmetic: bit-exact
    batches = [cref.plabel_stage_a(z, H, W) for z in zs]
    batches = [(mp, am.astype(np.int64)) for mp, am in batches]
    thr = {"CT": 0.9 * np.ones(C), "NT": None}.get(policy) if policy != "CBST" else ias_ref.cbst_threshold(batches, C, 0.2, 4)
    st = ias_ref.ConstantPolicyState(C, thr)
    want = np.concatenate([st.step(mp, am, ["img_%03d" % (t * B + b) for b in range(B)])
                           for t, (mp, am) in enumerate(batches)])
    assert np.array_equal(got, want)
    assert np.array_equal(np.load(os.path.join(root, "statics_class.npy")), st.statics_class)
    assert np.allclose(np.load(os.path.join(root, "class_mean_probabilities.npy")), st.class_mean_probs, rtol=1e-6)
    if policy == "NT":
        assert not os.path.exists(os.path.join(root, "class_threshold.npy")) and (got != 255).all()
    else:
        got_thr = np.load(os.path.join(root, "class_threshold.npy"))
        assert np.array_equal(got_thr.view(np.uint64), np.asarray(thr, np.float64).view(np.uint64))
    # (2) the reference's own artefacts (torch softmax): the same maps up to a few threshold-edge pixels
    tag = policy.lower()
    assert (got != g["plbl_" + tag]).mean() <= 2e-4
    assert np.abs(np.load(os.path.join(root, "statics_class.npy")) - g["statics_" + tag]).sum() <= 2e-4 * got.size
    assert np.allclose(np.load(os.path.join(root, "class_mean_probabilities.npy")), g["mean_" + tag], rtol=2e-4)
    if policy == "CBST":     # one fp16 bin (4.9e-4 below 1.0) either way
        assert np.abs(got_thr - g["thr_cbst"]).max() <= 1e-3
    stats = json.loads(open(os.path.join(root, "sample_class_stats.json")).read())
    assert len(stats) == nb * B and all("file" in s for s in stats)


# ------------------------------------------------------------------------------------------ K16 fused TTA
@pytest.mark.parametrize("flip", [False, True])
def test_tta_fused_vs_oracle_and_reference(golden, flip):
    from oracle import cref
    from hiast_amd import kernels as K
    from make_golden import TTA_SHAPE, TTA_SIZES, tta_inputs
    g = golden("tta")
    B, C, H, W = TTA_SHAPE
    t = tta_inputs()
    zs = [t[(hs, ws, False)] for hs, ws in TTA_SIZES]
    zfs = [t[(hs, ws, True)] for hs, ws in TTA_SIZES] if flip else None
    dz = [torch.from_numpy(z).cuda() for z in zs]
    dzf = [torch.from_numpy(z).cuda() for z in zfs] if flip else None
    probs, label = K.tta_fused(dz, dzf, TTA_SIZES, H, W, want_probs=True, want_label=True)
    oprobs, olabel = cref.tta(zs, zfs, TTA_SIZES, H, W)
    assert np.array_equal(probs.cpu().numpy().view(np.uint32), oprobs.view(np.uint32)), "probability sums differ from the oracle"
    assert np.array_equal(label.cpu().numpy(), olabel)
    _, only_label = K.tta_fused(dz, dzf, TTA_SIZES, H, W, want_probs=False, want_label=True)
    assert torch.equal(only_label, label)
    tag = "flip" if flip else "noflip"
    p = probs.cpu().numpy()
    assert np.abs(p[:, :, ::3, ::5] - g["probsum_" + tag]).max() <= 1e-5          # vs the reference's Validator
    top2 = np.sort(p, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-5
    assert np.array_equal(label.cpu().numpy()[clear], g["label_" + tag][clear])


def test_tta_fused_full_size_properties():
    """BASELINE-size geometry (2 x 19 x 512 x 1024, three scales incl. one above native, flip): the label map equals the
    argmax of the summed probabilities, every pixel's probabilities sum to (scales x 2), and a single native-size
    scale without flip reproduces pass 1's argmax map"""
    from hiast_amd import kernels as K
    B, C, H, W = 2, 19, 512, 1024
    sizes = [(384, 768), (512, 1024), (640, 1280)]
    zs = [torch.from_numpy(synth.smooth_logits_lr(3300 + i, B, C, s[0] // 8, s[1] // 8)).cuda() for i, s in enumerate(sizes)]
    zfs = [torch.from_numpy(synth.smooth_logits_lr(3310 + i, B, C, s[0] // 8, s[1] // 8)).cuda() for i, s in enumerate(sizes)]
    probs, label = K.tta_fused(zs, zfs, sizes, H, W, want_probs=True, want_label=True)
    top2 = probs.topk(2, dim=1).values
    clear = top2[:, 0] > top2[:, 1]                                     # (exact ties: the kernel takes the first class)
    assert torch.equal(probs.argmax(1).to(torch.uint8)[clear], label[clear]) and float(clear.float().mean()) > 0.9999
    assert float((probs.sum(1) - 2 * len(sizes)).abs().max()) <= 2e-5
    _, lab1 = K.tta_fused(zs[1:2], None, sizes[1:2], H, W)
    _, am, _ = K.plabel_pass1(zs[1], H, W)
    # (a logit within 6e-8 of the maximum has exp() == 1.0: the probability argmax may then name the earlier class)
    assert int((lab1 != am).sum()) <= 4


def test_validator_tta_on_hip_matches_oracle(tmp_path):
    """validate.py with resize_sizes x is_flip on the device (fused tail) vs the oracle end to end: oracle forward
    (torch CPU, fp32) -> orc_tta -> IoU: mIoU within 0.05 points, label maps equal on > 99.9 % of the pixels"""
    from oracle import cref, deeplab_ref, metrics_ref
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import MODEL
    from hiast_amd.tools import synth_data
    from hiast_amd.workflows.validator import Validator
    from make_golden import seeded_state_dict
    H, W, C = 128, 256, 19
    c = synth_data.synthetic_cfg(str(tmp_path), n_train=1, n_val=4, h=H, w=W)
    c.model.type = "SourceOnlySegmentor"
    m = MODEL["SourceOnlySegmentor"](c)
    sd = {"seg_model." + k: v for k, v in seeded_state_dict(m.seg_model, 4343).items()}
    for i in range(4):
        sd["seg_model.aspp.conv2d_list.%d.weight" % i] *= 30.0
    ck = str(tmp_path / "ck.pth")
    torch.save(sd, ck)
    c.validate.resume_from = ck
    c.validate.resize_sizes = [[96, 192], [128, 256], [160, 320]]
    c.validate.is_flip = True
    c.validate.color_mask_dir_path = str(tmp_path / "masks")
    v = Validator(c, device=torch.device("cuda", 0))
    miou = v.run()
    F = torch.nn.functional
    inter, union, agree, total = np.zeros(C, np.int64), np.zeros(C, np.int64), 0, 0
    torch.set_num_threads(16)
    from PIL import Image
    from hiast_amd.sseg.datasets import utils as du
    for data in v.v_loader:
        imgs = data["images"]
        if imgs.dtype == torch.uint8:      # the validator normalises on the device (round 6); the oracle takes the HOST transform
            imgs = torch.stack([du._img_to_tensor(i.numpy(), du.MEAN, du.STD) for i in imgs])
        zs, zfs = [], []
        with torch.no_grad():
            for size in c.validate.resize_sizes:
                x = F.interpolate(imgs, size, mode="bilinear", align_corners=True)
                zs.append(deeplab_ref.segmentor_logits(x, sd)[1].numpy())
                zfs.append(deeplab_ref.segmentor_logits(torch.flip(x, dims=[3]), sd)[1].numpy())
        _, lab = cref.tta(zs, zfs, c.validate.resize_sizes, H, W, want_probs=False)
        a, b = metrics_ref.intersection_and_union(lab.astype(np.int64), data["labels"].numpy().astype(np.int64), C)
        inter += a
        union += b
        for k, p in enumerate(data["image_paths"]):
            mask = np.array(Image.open(os.path.join(c.validate.color_mask_dir_path, os.path.basename(p))))
            agree += int((mask == lab[k]).sum())
            total += mask.size
    want, _, _ = metrics_ref.miou(inter.astype(np.float64), union.astype(np.float64))
    assert abs(100 * miou - 100 * want) <= 0.05, (miou, want)
    assert agree / total >= 0.999, agree / total
