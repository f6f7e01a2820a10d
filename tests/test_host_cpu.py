"""Host-side logic on CPU (no GPU): config tree, registries, model surface / state-dict keys / CPU
forward against the reference golden, datasets + transform, CopyPaste golden, generator host logic with
an oracle-backed engine, Validator on CPU (BASELINE config 1)."""
import json
import os

import numpy as np
import pytest
import torch

import synth
from hiast_amd.utils.default_config import get_default_cfg, CfgNode
from hiast_amd.utils.registry import register  # noqa: F401
from hiast_amd.utils.registry.registries import (DATASET, LOSS, MODEL, PREPROCESSOR, PSEUDO_POLICY, SEG_MODEL,
                                                 TRAINER)
from hiast_amd.tools import synth_data


def test_registries_hold_reference_names():
    assert {"CE", "SoftCE", "MSE", "KLDIV", "BCEWithLogits"} <= set(LOSS)
    assert {"Cityscapes", "GTAV", "SYNTHIA", "Oxford"} <= set(DATASET)
    assert {"SelfTrainingSegmentor", "SourceOnlySegmentor", "AdversarialWarmupSegmentor"} <= set(MODEL)
    assert {"SelfTrainingTrainer", "ConsistencySelfTrainingTrainer", "SourceOnlyTrainer",
            "AdversarialWarmupTrainer"} <= set(TRAINER)
    assert {"IAS", "CT", "NT", "CBST"} <= set(PSEUDO_POLICY)
    assert "CopyPaste" in PREPROCESSOR and "DeepLab_V2" in SEG_MODEL
    with pytest.raises(AssertionError):
        SEG_MODEL.register("DeepLab_V2", object)


def test_cfg_yaml_semantics(tmp_path):
    c = get_default_cfg()
    y = tmp_path / "a.yaml"
    y.write_text("train:\n  lr: 3e-6\n  batch_size: 6\nvalidate:\n  color_mask_dir_path: None\n"
                 "dataset:\n  target:\n    aug_type: [ 'MS', 'CCA' ]\n")
    c.merge_from_file(str(y))
    assert c.train.lr == 3e-6 and isinstance(c.train.lr, float) and c.train.batch_size == 6
    assert c.validate.color_mask_dir_path is None and c.dataset.target.aug_type == ["MS", "CCA"]
    bad = tmp_path / "b.yaml"
    bad.write_text("train:\n  no_such_key: 1\n")
    with pytest.raises(KeyError):
        c.merge_from_file(str(bad))
    c.freeze()
    with pytest.raises(AttributeError):
        c.train.lr = 1.0
    assert isinstance(CfgNode(json.loads(json.dumps(c.to_dict()))), CfgNode)
    assert "lr: 3.0e-06" in c.dump() or "lr: 3e-06" in c.dump()


@pytest.mark.needs_reference
def test_reference_yaml_files_parse_unchanged():
    for f in ("sl_1.yaml", "sl_2.yaml", "sl_3.yaml"):
        c = get_default_cfg()
        c.merge_from_file("/root/reference/code/configs/" + f)
        c.merge_from_file("/root/reference/code/configs/hiast_setting.yaml")
        assert c.trainer == "ConsistencySelfTrainingTrainer" and c.pseudo_policy.type == "IAS"
    c = get_default_cfg()
    c.merge_from_file("/root/reference/code/configs/validate.yaml")
    assert c.model.type == "SourceOnlySegmentor" and c.validate.resize_sizes == [[768, 1536]]


def test_model_cpu_forward_matches_reference_golden(golden):
    from make_golden import seeded_state_dict
    g = golden("deeplab")
    m = SEG_MODEL["DeepLab_V2"](num_classes=19, output_dim=256)
    assert list(m.state_dict().keys()) == json.loads(str(g["keys"]))
    m.load_state_dict(seeded_state_dict(m, 9000))
    m.eval()
    torch.set_num_threads(8)
    x = torch.from_numpy(synth.normal_f32(900 + ord("a"), (1, 3, 65, 129)))
    with torch.no_grad():
        pred, feat = m(x)
    assert np.allclose(pred.numpy(), g["pred_a"], rtol=1e-4, atol=1e-4)
    assert np.allclose(feat.numpy()[:, ::64], g["feat_sub_a"], rtol=1e-4, atol=1e-4)
    groups = m.get_optimizer_params(1e-4)
    assert [gr["lr"] for gr in groups] == [1e-4, 1e-3, 1e-3]
    n_params = sum(p.numel() for p in m.parameters())
    assert n_params == 44425612


def test_segmentor_surface_cpu():
    c = get_default_cfg()
    c.model.type = "SelfTrainingSegmentor"
    seg = MODEL["SelfTrainingSegmentor"](c).eval()
    assert all(k.startswith("seg_model.") for k in seg.state_dict())
    x = torch.zeros(1, 3, 33, 65)
    with torch.no_grad():
        out = seg(x)
        lo = seg(x, lowres=True)
    assert tuple(out["logits"].shape) == (1, 19, 33, 65) and tuple(out["backbone"].shape) == (1, 2048, 5, 9)
    assert tuple(lo["logits_lowres"].shape) == (1, 19, 5, 9) and lo["size"] == (33, 65)
    with pytest.raises(RuntimeError):      # losses are HIP-only: fail loudly on CPU tensors
        seg.compute_loss(out["logits"], torch.zeros(1, 33, 65, dtype=torch.long))


def test_dataset_transform_and_pseudo_paths(tmp_path):
    c = synth_data.synthetic_cfg(str(tmp_path), n_train=3, n_val=2, h=32, w=64)
    ds = DATASET["Cityscapes"](c, c.dataset.target.json_path, c.dataset.target.image_dir, aug_type=["PRS-16-32"])
    it = ds[1]
    assert set(it) == {"images", "labels", "image_paths"}
    assert it["images"].dtype == torch.float32 and tuple(it["images"].shape) == (3, 16, 32)
    assert it["labels"].dtype == torch.int64 and tuple(it["labels"].shape) == (16, 32)
    img, lbl, path = ds.load_data(1)
    from PIL import Image
    want = np.asarray(Image.fromarray(img).resize((32, 16), Image.BILINEAR)).astype(np.float32) / 255
    want = (want - np.array([0.485, 0.456, 0.406], np.float32)) / np.array([0.229, 0.224, 0.225], np.float32)
    assert np.allclose(it["images"].numpy(), want.transpose(2, 0, 1), atol=1e-6)
    assert path.split("/")[-1].endswith("_leftImg8bit.png") and len(path.split("/")) >= 5
    two = DATASET["Cityscapes"](c, c.dataset.target.json_path, c.dataset.target.image_dir, aug_type=["MS", "CCA"])
    it2 = two[0]
    assert isinstance(it2["images"], list) and len(it2["images"]) == 2
    assert tuple(it2["images"][0].shape) == (3, 512, 1024) and torch.equal(it2["labels"][0], it2["labels"][1])


def test_oxford_labels_and_warmup_optimizers(tmp_path):
    """Oxford RobotCar: RGBA label PNGs whose first channel is the class id (oxford_dataset.py:14-23); the
    discriminator gets its own Adam(lr = discriminator.lr) (utils.py:148-152) and scheduler (:157-163)."""
    from PIL import Image
    from hiast_amd.utils import utils
    from hiast_amd.sseg.datasets.loader.oxford_dataset import OxfordDataset
    ids = np.array([[0, 1, 2, 3, 4, 5, 6, 7], [8, 9, 10, 11, 12, 13, 14, 17]], np.uint8)
    rgba = np.stack([ids, ids * 0 + 9, ids * 0 + 9, ids * 0 + 255], -1)
    path = str(tmp_path / "lbl.png")
    Image.fromarray(rgba, "RGBA").save(path)
    ds = object.__new__(OxfordDataset)
    ds.num_classes = 9
    got = ds.read_label(path)
    want = np.array([[255, 0, 1, 2, 3, 4, 5, 6], [255, 255, 7, 8, 8, 8, 8, 8]], np.uint8)
    assert np.array_equal(got, want)
    assert ds.read_label(str(tmp_path / "frame.jpg")) is None          # unlabeled training frames
    c = synth_data.synthetic_cfg(str(tmp_path), n_train=1, n_val=1, h=16, w=32)
    c.model.type = "AdversarialWarmupSegmentor"
    c.model.discriminator.is_enabled = True
    c.train.total_iter = 10
    m = MODEL[c.model.type](c)
    assert [k for k in m.state_dict() if k.startswith("D.")] == [
        "D.%s.%s" % (n, t) for n in ("conv1", "conv2", "conv3", "conv4", "classifier") for t in ("weight", "bias")]
    g_opt, d_opt = utils.init_optimizers(c, m)
    assert len(g_opt.param_groups) == 3 and len(d_opt.param_groups) == 1
    assert d_opt.param_groups[0]["lr"] == c.model.discriminator.lr and d_opt.param_groups[0]["betas"] == (0.9, 0.999)
    assert sum(p.numel() for p in d_opt.param_groups[0]["params"]) == sum(p.numel() for p in m.D.parameters())
    assert len(utils.init_schedulers(c, g_opt, d_opt)) == 2


def test_copy_paste_rejection_sampling_is_bounded():
    """nearly all sampling mass on classes no pseudo-labelled image contains (a barely trained model): the class draw
    must still return promptly, and only usable hard classes come out"""
    import types
    from hiast_amd.sseg.datasets.preprocessor import CopyPaste
    cp = object.__new__(CopyPaste)
    cp.cfg = types.SimpleNamespace(dataset=types.SimpleNamespace(num_classes=19))
    cp.samples_with_class = {c: (["a.png"] if c in (0, 1, 5, 17) else []) for c in range(19)}
    v = np.zeros(19)
    v[[0, 1, 5, 17]] = 1.0 - 1e-6
    cp.class_value = v
    cp.class_probs = CopyPaste.calculate_class_probs(cp)
    np.random.seed(3)
    got = {int(cp.random_select(list(range(14)))) for _ in range(50)}
    assert got <= {0, 1, 5} and len(got) >= 2          # 17 is outside the selected (hard) set
    cp.samples_with_class = {c: [] for c in range(19)}
    assert cp.random_select(list(range(14))) is None


def test_copy_paste_matches_reference_golden(golden):
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
    assert np.array_equal(cp.hard_classes, g["hard_classes"])
    assert np.allclose(cp.class_probs, g["class_probs"], rtol=1e-12)
    np.random.seed(888)
    for i in range(N):
        im, lb, mk = cp.run(imgs[i].copy(), lbls[i].copy())
        assert np.array_equal(im, g["img"][i]) and np.array_equal(lb, g["lbl"][i]) and np.array_equal(mk, g["mask"][i])
    # SYNTHIA: absent classes get probability 0 instead of the reference's NaN
    c2 = get_default_cfg()
    c2.dataset.source.type = "SYNTHIA"
    cp2 = PREPROCESSOR["CopyPaste"](c2, DS(), g["class_value"].copy())
    assert np.isfinite(cp2.class_probs).all() and cp2.class_probs[[9, 14, 16]].sum() == 0
    assert not set(cp2.hard_classes) & {9, 14, 16}


class OracleEngine:
    """test double with the HIP engine's interface, backed by the CPU oracle: lets the generator's
    host logic (ordering, sharding, statistics, artefacts) run without a GPU"""

    def __init__(self, C, H, W):
        from oracle import cref
        self.cref, self.C, self.H, self.W = cref, C, H, W
        self.device = torch.device("cpu")

    def pass1(self, imgs):
        from hiast_amd.workflows import ias_math
        if imgs is None or imgs.shape[0] == 0:
            self.mp = None
            return torch.zeros((self.C, ias_math.NBINS), dtype=torch.int32)
        # "model": low-res logits = a fixed function of the image tensor
        z = torch.nn.functional.adaptive_avg_pool2d(imgs, (self.H // 8, self.W // 8))
        z = torch.cat([z * (k + 1) for k in range(7)], 1)[:, :self.C].contiguous().numpy() * 3
        self.mp, self.am = self.cref.plabel_stage_a(z, self.H, self.W)
        return torch.from_numpy(self.cref.plabel_hist(self.mp, self.am, self.C).view(np.int32))

    def strided_hist(self, interval, rank_offset=None):
        """the reference's own formulation: tmp = probs[lbls == c].astype(f16); tmp[first::interval]"""
        from hiast_amd.workflows import ias_math
        hist = np.zeros((self.C, ias_math.NBINS), np.int32)
        if self.mp is None:
            return torch.from_numpy(hist)
        off = np.zeros(self.C, np.int64) if rank_offset is None else np.asarray(rank_offset, np.int64)
        for c in range(self.C):
            tmp = self.mp[self.am == c].astype(np.float16)
            first = (-int(off[c])) % interval
            np.add.at(hist[c], tmp[first::interval].view(np.uint16).astype(np.int64), 1)
        return torch.from_numpy(hist)

    def pass2(self, thr):
        if self.mp is None:
            return None, torch.zeros((0, self.C), dtype=torch.int64), torch.zeros(self.C, dtype=torch.int64)
        plbl, count, sfx = self.cref.plabel_select(self.mp, self.am, thr, self.C)
        return torch.from_numpy(plbl), torch.from_numpy(count), torch.from_numpy(sfx.view(np.int64))


class PipelinedOracleEngine(OracleEngine):
    """the same double with the HIP engine's begin() / hist_host() / finish() interface: the generators then run their
    software-pipelined loop (forward + pass 1 of batch t+1 before the histogram of batch t is used)"""

    def begin(self, imgs):
        hist = self.pass1(imgs)
        st = {"hist": hist, "mp": self.mp, "am": None if self.mp is None else self.am}
        self.mp = self.am = "consumed"          # a state must carry everything pass 2 needs
        return st

    def post_stream(self):
        return None

    def hist_host(self, st, allreduce=None):
        h = st["hist"] if allreduce is None else allreduce(st["hist"])
        return h.numpy().view(np.uint32)

    def finish(self, st, thr):
        self.mp, self.am = st["mp"], st["am"]
        try:
            return self.pass2(thr)
        finally:
            self.mp = self.am = "consumed"


@pytest.mark.parametrize("policy", ["IAS", "CT", "CBST"])
def test_pipelined_generator_loop_writes_the_same_artefacts(tmp_path, policy):
    """the software-pipelined batch loop (begin / hist_host / finish engines) against the one-batch-at-a-time loop"""
    from PIL import Image
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import PSEUDO_POLICY
    from hiast_amd.tools import synth_data
    h, w, C = 32, 64, 19
    c = synth_data.synthetic_cfg(str(tmp_path), n_train=7, n_val=1, h=h, w=w)
    c.pseudo_policy.batch_size = 2
    c.pseudo_policy.cbst.sample_interval = 3
    out = {}
    for name, eng in (("serial", OracleEngine(C, h, w)), ("pipelined", PipelinedOracleEngine(C, h, w))):
        c.pseudo_policy.save_dir = os.path.join(str(tmp_path), name, "pseudo_labels")
        gen = PSEUDO_POLICY[policy](c, engine=eng)
        gen.run()
        d = c.pseudo_policy.save_dir
        out[name] = ({f: np.load(os.path.join(d, "..", f)) for f in ("statics_class.npy", "class_mean_probabilities.npy")},
                     {n: np.array(Image.open(os.path.join(d, n))) for n in sorted(os.listdir(d))},
                     None if gen.class_threshold is None else np.asarray(gen.class_threshold).view(np.uint64).copy())
    a, b = out["serial"], out["pipelined"]
    assert len(a[1]) == 7 and a[1].keys() == b[1].keys()
    assert all(np.array_equal(a[0][f], b[0][f]) for f in a[0]) and all(np.array_equal(a[1][n], b[1][n]) for n in a[1])
    assert (a[2] is None and b[2] is None) or np.array_equal(a[2], b[2])


def test_generator_host_logic_and_artefacts(tmp_path):
    """IAS generator end to end on CPU with the oracle engine: artefact files, formats, and equality
    with the oracle's list/quantile formulation driven in the same image order."""
    from oracle import ias_ref
    from PIL import Image
    h, w, C = 32, 64, 19
    c = synth_data.synthetic_cfg(str(tmp_path), n_train=5, n_val=1, h=h, w=w)
    c.pseudo_policy.batch_size = 2
    eng = OracleEngine(C, h, w)
    gen = PSEUDO_POLICY["IAS"](c, engine=eng)
    gen.run()
    root = os.path.join(c.pseudo_policy.save_dir, "..")
    thr = np.load(os.path.join(root, "class_threshold.npy"))
    stats = np.load(os.path.join(root, "statics_class.npy"))
    means = np.load(os.path.join(root, "class_mean_probabilities.npy"))
    sample_stats = json.load(open(os.path.join(root, "sample_class_stats.json")))
    swc = json.load(open(os.path.join(root, "samples_with_class.json")))
    assert thr.dtype == np.float64 and thr.shape == (C,) and stats.dtype == np.int64 and means.shape == (C,)
    assert len(sample_stats) == 5 and set(swc) == {str(i) for i in range(C)}
    # replay with the oracle in the generator's (seeded, shuffled) order
    st = ias_ref.IASState(C, c.pseudo_policy.ias.alpha, c.pseudo_policy.ias.beta, c.pseudo_policy.ias.gamma, 0.99)
    eng2 = OracleEngine(C, h, w)
    for data in gen.t_loader:
        eng2.pass1(data["images"])
        plbl = st.step(eng2.mp, eng2.am.astype(np.int64), data["image_paths"])
        for b, p in enumerate(data["image_paths"]):
            name = os.path.splitext(os.path.basename(p))[0] + "_pseudo_label.png"
            got = np.array(Image.open(os.path.join(c.pseudo_policy.save_dir, name)))
            assert got.dtype == np.uint8 and np.array_equal(got, plbl[b])
    assert np.array_equal(thr.view(np.uint64), st.class_threshold.view(np.uint64))
    assert np.array_equal(stats, st.statics_class)
    assert np.allclose(means, st.class_mean_probs, rtol=1e-6)
    assert sample_stats == json.loads(json.dumps(st.sample_stats))
    # the dataset side consumes these artefacts (CopyPaste input)
    ds = DATASET["Cityscapes"](c, c.dataset.target.json_path, c.dataset.target.image_dir,
                               pseudo_dir=c.pseudo_policy.save_dir, aug_type=[])
    assert set(ds.get_samples_with_class()) == set(range(C))
    img, lbl, _ = ds.load_data(0)
    assert lbl.dtype == np.uint8 and lbl.shape == (h, w)


def test_validator_cpu_config1(tmp_path):
    """BASELINE config 1: validate.py, CPU only, SourceOnlySegmentor, 4 synthetic 512x256 images;
    mIoU equals the oracle's (functional torch restatement + C IoU counts)."""
    from oracle import deeplab_ref, metrics_ref
    from hiast_amd.workflows.validator import Validator
    from make_golden import seeded_state_dict
    H, W = 64, 128        # small here; the full 256x512 case runs in bench.py's cpu leg
    c = synth_data.synthetic_cfg(str(tmp_path), n_train=1, n_val=4, h=H, w=W)
    c.model.type = "SourceOnlySegmentor"
    m = MODEL["SourceOnlySegmentor"](c)
    sd = {"seg_model." + k: v for k, v in seeded_state_dict(m.seg_model, 4242).items()}
    ck = tmp_path / "ck.pth"
    torch.save(sd, str(ck))
    c.validate.resume_from = str(ck)
    torch.set_num_threads(8)
    v = Validator(c, device=torch.device("cpu"))
    miou = v.run()
    inter = np.zeros(19, np.int64)
    union = np.zeros(19, np.int64)
    for data in v.v_loader:
        with torch.no_grad():
            logits, _, _ = deeplab_ref.segmentor_logits(data["images"], sd)
        pred = torch.softmax(logits, 1).argmax(1).numpy()
        i, u = metrics_ref.intersection_and_union(pred, data["labels"].numpy(), 19)
        inter += i
        union += u
    want, _, _ = metrics_ref.miou(inter, union)
    assert abs(miou - want) <= 0.05 / 100 + 1e-12


def test_bench_roofline_entries_name_the_binding_roof():
    """bench.py's roofline bookkeeping (pure host arithmetic): the 3x3 split-plane launch is priced against the dense
    bf16 MFMA peak / 3, the 256->1024 1x1 (+residual) launch against the 8 TB/s HBM roof; achieved = algorithmic work of
    one launch / its average duration"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    # ("igemm", x [B,H,W,PL*Cin], wp [Cout,taps,PL*Cin], PL, stride, dil, has_res, out_f32, has_bn, relu)
    k3 = ("igemm", (8, 64, 128, 512), (256, 9, 512), 2, 1, 2, False, False, True, True)
    r = bench.roofline_of(k3, 0.18, 22, 1)
    flop = 2.0 * 8 * 64 * 128 * 9 * 256 * 256
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["peak"] - 2500.0 / 3) < 1e-9
    assert abs(r["achieved"] - flop / 0.18e-3 / 1e12) < 1e-6 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["launches_per_step"] == 22
    k1 = ("igemm", (8, 64, 128, 512), (1024, 1, 512), 2, 1, 1, True, False, True, True)
    r = bench.roofline_of(k1, 0.17, 23, 1)
    M = 8 * 64 * 128
    alg = (M * 512 + 1024 * 512) * 2 + M * 1024 * 4 * 2            # input + weights + (residual + output) hi|lo pairs
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["achieved"] - alg / 0.17e-3 / 1e9) < 1e-3 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    # the pure host path of the step split: the serial switch exists and the JSON contract fields are spelled as the driver reads them
    src = open(os.path.join(os.path.dirname(__file__), "..", "bench.py")).read()
    for field in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"', '"higher_is_better"',
                  '"scaling"', '"vs_baseline"', '"dtype"', '"data"', '"config"', '"roofline"', '"cpu_baseline"'):
        assert field in src, field


def test_decoded_cache_returns_the_decoded_bytes(tmp_path):
    """cfg.dataset.decoded_cache_dir: load_data gives the same arrays with and without the cache and on the second
    (cached) read; entries are private writable copies, follow the source file (a rewritten file is a new entry),
    different label id maps do not collide, and the size bound is respected"""
    from PIL import Image
    from hiast_amd.sseg.datasets.decoded_cache import DecodedCache
    c = synth_data.synthetic_cfg(str(tmp_path / "data"), n_train=3, n_val=2, h=32, w=64)
    plain = DATASET["Cityscapes"](c, c.dataset.target.json_path, c.dataset.target.image_dir)
    want = [plain.load_data(i) for i in range(3)]
    c.dataset.decoded_cache_dir = str(tmp_path / "cache")
    ds = DATASET["Cityscapes"](c, c.dataset.target.json_path, c.dataset.target.image_dir)
    for rnd in range(2):                                    # first round decodes and stores, second reads the cache
        for i in range(3):
            img, lbl, path = ds.load_data(i)
            assert np.array_equal(img, want[i][0]) and np.array_equal(lbl, want[i][1]) and path == want[i][2]
            assert img.flags.writeable and lbl.flags.writeable and img.dtype == np.uint8
            img[:] = 0                                      # a private copy: the stored entry is not touched
        assert len([f for f in os.listdir(c.dataset.decoded_cache_dir) if f.endswith(".npy")]) == 6
    nine = DATASET["Cityscapes"](c, c.dataset.target.json_path, c.dataset.target.image_dir, num_classes=9)
    c9 = get_default_cfg()
    assert np.array_equal(nine.load_data(0)[1], nine.read_label(nine.lbl_path_list[0]))     # 9-class map: own entry
    assert not np.array_equal(nine.load_data(0)[1], want[0][1])
    # a rewritten source file is a new entry
    p = ds.img_path_list[0]
    new = np.full((32, 64, 3), 7, dtype=np.uint8)
    st = os.stat(p)
    Image.fromarray(new).save(p)
    os.utime(p, ns=(st.st_atime_ns, st.st_mtime_ns + 10 ** 9))
    assert np.array_equal(ds.load_data(0)[0], new)
    # size bound: nothing is written beyond it
    small = DecodedCache(str(tmp_path / "small"), max_gb=1e-6)
    assert small.load(p, lambda q: np.array(Image.open(q).convert("RGB"), dtype=np.uint8)).shape == (32, 64, 3)
    assert os.listdir(str(tmp_path / "small")) == []
    assert c9.dataset.decoded_cache_dir is None             # off unless asked for (the reference has no such feature)


def test_set_mode_looks_at_the_wrapped_net():
    from hiast_amd.utils import utils
    from hiast_amd.workflows.trainer.base_trainer import _Bare
    net = torch.nn.Sequential(torch.nn.BatchNorm2d(4))
    wrap = _Bare(net)
    net.eval()                      # wrapper still says training=True
    utils.set_mode(wrap, True)
    assert wrap.training and net.training and net[0].training
    utils.set_mode(wrap, False)
    assert not wrap.training and not net[0].training


def test_limit_cpu_threads_caps_and_respects_override(monkeypatch):
    """utils.limit_cpu_threads: torch's intra-op pool is capped (never raised), HIAST_CPU_THREADS overrides, 0 leaves it"""
    import torch
    from hiast_amd.utils import utils
    before = torch.get_num_threads()
    try:
        torch.set_num_threads(6)
        monkeypatch.setenv("HIAST_CPU_THREADS", "0")
        utils.limit_cpu_threads()
        assert torch.get_num_threads() == 6
        monkeypatch.setenv("HIAST_CPU_THREADS", "8")
        utils.limit_cpu_threads()
        assert torch.get_num_threads() == 6            # a cap, not a setting
        monkeypatch.delenv("HIAST_CPU_THREADS")
        utils.limit_cpu_threads()
        assert torch.get_num_threads() == 4
        monkeypatch.setenv("HIAST_CPU_THREADS", "2")
        utils.limit_cpu_threads()
        assert torch.get_num_threads() == 2
    finally:
        torch.set_num_threads(before)


def test_training_loader_chains_epochs_like_restarted_iterators():
    """BaseTrainer's training DataLoader walks on into the next epoch without a new iterator (_EpochChain); the batches
    are those of the reference's loop: iterator per epoch, set_epoch(epoch + 1) on StopIteration, incomplete batch dropped."""
    import torch
    from torch.utils.data import DataLoader, DistributedSampler
    from hiast_amd.workflows.trainer.base_trainer import _EpochChain
    ds = list(range(23))
    ref_s = DistributedSampler(ds, num_replicas=2, rank=1, shuffle=True)
    ref_l = DataLoader(ds, 4, sampler=ref_s, drop_last=True)
    want, it = [], iter(ref_l)
    while len(want) < 9:
        try:
            want.append(next(it).tolist())
        except StopIteration:
            ref_s.set_epoch(ref_s.epoch + 1)
            it = iter(ref_l)
    s = DistributedSampler(ds, num_replicas=2, rank=1, shuffle=True)
    loader = DataLoader(ds, batch_sampler=_EpochChain(s, 4, True))
    assert len(loader) == len(ref_l) == 3
    it = iter(loader)
    got = [next(it).tolist() for _ in range(9)]
    assert got == want and s.epoch >= 2
    assert want[0] != want[3]                       # the epochs are shuffled differently
    # a dataset smaller than one batch ends instead of spinning
    tiny = DistributedSampler(list(range(3)), num_replicas=1, rank=0, shuffle=True)
    assert list(_EpochChain(tiny, 4, True)) == []
