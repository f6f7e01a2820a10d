"""End-to-end on the MI355X with the real HIP engine, on a synthetic Cityscapes-layout dataset:
config 2 (IAS pseudo-label generation) -> config 3 (a few HIAST self-training iterations incl. CopyPaste, EMA,
validation, checkpoints) -> validate.py.  The generator's artefacts are replayed against the oracle
(bit-exact label maps / thresholds given the same low-res logits); mIoU is compared with the oracle's."""
import json
import os

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu

H, W, C = 128, 256, 19


@pytest.fixture(scope="module")
def world(tmp_path_factory):
    assert torch.cuda.is_available()
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import MODEL
    from hiast_amd.tools import synth_data
    from make_golden import seeded_state_dict
    root = str(tmp_path_factory.mktemp("e2e"))
    cfg = synth_data.synthetic_cfg(root, n_train=6, n_val=4, h=H, w=W)
    # the 2-3 iteration plumbing tests below use the 16-bit type WITHOUT loss scaling: under fp16 (the default, apex O1) the
    # first ~10 iterations of a run are skipped while the dynamic scale comes down from 2^16, as under apex — covered by
    # tests/test_gpu_fp16.py::test_fp16_training_step_runs_the_trunk_on_the_own_kernels
    cfg.train.amp_dtype = "bf16"
    m = MODEL["SelfTrainingSegmentor"](cfg)
    sd = {"seg_model." + k: v for k, v in seeded_state_dict(m.seg_model, 777).items()}
    # the state of a trained checkpoint as far as ranges go: running statistics = those of the data (an eval forward on
    # statistics that do not belong to the weights grows block by block — past fp16's range), head calibrated so that
    # max-probs are spread over (0.1, 1): logits std ~ 3
    m.load_state_dict(sd)
    m = m.cuda().eval()
    ds = np.stack([synth_data.make_sample(5 + i, H, W)[0].astype(np.float32).transpose(2, 0, 1) for i in range(2)]) / 255.0
    xs = torch.from_numpy((ds - 0.45) / 0.225).cuda()
    synth_data.calibrate_bn(m, xs)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        z = m(xs[:1], lowres=True)["logits_lowres"]
    scale = 3.0 / float(z.std())
    for i in range(4):
        sd["seg_model.aspp.conv2d_list.%d.weight" % i] = sd["seg_model.aspp.conv2d_list.%d.weight" % i] * scale
        sd["seg_model.aspp.conv2d_list.%d.bias" % i] = sd["seg_model.aspp.conv2d_list.%d.bias" % i] * scale
    del m
    ck = os.path.join(root, "warmup.pth")
    torch.save(sd, ck)
    cfg.pseudo_policy.resume_from = ck
    cfg.train.resume_from = ck
    cfg.validate.resume_from = ck
    return cfg, sd, root


def test_config2_generator_matches_oracle(world):
    from PIL import Image
    from oracle import cref, ias_ref
    from hiast_amd.utils.registry.registries import PSEUDO_POLICY
    cfg, sd, root = world
    gen = PSEUDO_POLICY["IAS"](cfg)
    gen.run()
    pdir = cfg.pseudo_policy.save_dir
    assert len(os.listdir(pdir)) == 6
    thr = np.load(os.path.join(pdir, "..", "class_threshold.npy"))
    stats = np.load(os.path.join(pdir, "..", "statics_class.npy"))
    # replay: same model forward on the device for the low-res logits, then the ORACLE for everything else
    st = ias_ref.IASState(C, cfg.pseudo_policy.ias.alpha, cfg.pseudo_policy.ias.beta, cfg.pseudo_policy.ias.gamma, 0.99)
    model = gen.engine.model
    from hiast_amd.sseg.datasets import utils as du
    for data in gen.t_loader:
        imgs = data["images"]
        if imgs.dtype == torch.uint8:      # the generator normalises on the device; the replay uses the HOST transform
            imgs = torch.stack([du._img_to_tensor(i.numpy(), du.MEAN, du.STD) for i in imgs])
        with torch.no_grad():
            z = model(imgs.cuda(), lowres=True)["logits_lowres"].float().cpu().numpy()
        mp, am = cref.plabel_stage_a(z, H, W)
        plbl = st.step(mp, am.astype(np.int64), data["image_paths"])
        for b, p in enumerate(data["image_paths"]):
            name = os.path.splitext(os.path.basename(p))[0] + "_pseudo_label.png"
            got = np.array(Image.open(os.path.join(pdir, name)))
            assert np.array_equal(got, plbl[b]), name
    assert np.array_equal(thr.view(np.uint64), st.class_threshold.view(np.uint64))
    assert np.array_equal(stats, st.statics_class)
    assert stats.sum() > 0, "the synthetic warm-up model should keep some pixels"
    means = np.load(os.path.join(pdir, "..", "class_mean_probabilities.npy"))
    assert np.allclose(means, st.class_mean_probs, rtol=1e-6)


def test_config3_hiast_training_round(world):
    """ConsistencySelfTrainingTrainer for a few iterations on the labels of config 2."""
    from hiast_amd.utils.registry.registries import TRAINER
    cfg, sd, root = world
    c = cfg.clone()
    c.trainer = "ConsistencySelfTrainingTrainer"
    c.dataset.target.pseudo_dir = cfg.pseudo_policy.save_dir
    c.dataset.target.aug_type = ["PRS-%d-%d" % (H, W), "CCA"]      # small views: 'MS' always crops to 512x1024
    c.cst_training.is_enabled = True
    c.cst_training.cst_loss.weight = 0.5
    c.preprocessor.type = "CopyPaste"
    c.train.gpu_num = 1
    c.train.batch_size = 2
    c.train.total_iter = 2
    c.train.iter_report = 1
    c.train.iter_val = 2
    c.train.lr = 3e-6
    c.work_dir = os.path.join(root, "work_hiast")
    c.freeze()
    tr = TRAINER[c.trainer](c, 0)
    p0 = next(tr.model.module.seg_model.aspp.parameters()).detach().clone()
    e0 = next(tr.ema_model.seg_model.aspp.parameters()).detach().clone()
    losses = tr.train()
    assert set(losses) == {"target_seg_loss", "kld_confident_loss", "ent_ignored_loss", "cst_loss"}
    assert all(torch.isfinite(v) for v in losses.values()), losses
    tr.run()
    p1 = next(tr.model.module.seg_model.aspp.parameters()).detach()
    e1 = next(tr.ema_model.seg_model.aspp.parameters()).detach()
    assert not torch.equal(p0, p1), "student did not move"
    assert not torch.equal(e0, e1) and (e1 - e0).abs().max() < (p1 - p0).abs().max(), "EMA teacher must lag"
    ck = os.path.join(c.work_dir, "checkpoints")
    assert {"model_last.pth", "ema_model_last.pth"} <= set(os.listdir(ck))
    saved = torch.load(os.path.join(ck, "model_last.pth"), map_location="cpu")
    assert list(saved.keys()) == list(sd.keys())          # reference state-dict keys, no 'module.' prefix


def test_config3_plain_self_training_trainer(world):
    from hiast_amd.utils.registry.registries import TRAINER
    cfg, sd, root = world
    c = cfg.clone()
    c.trainer = "SelfTrainingTrainer"
    c.dataset.target.pseudo_dir = cfg.pseudo_policy.save_dir
    c.dataset.target.aug_type = ["PRS-%d-%d" % (H, W)]
    c.train.gpu_num = 1
    c.train.batch_size = 2
    c.train.total_iter = 2
    c.train.iter_report = 1
    c.train.iter_val = 100
    c.work_dir = os.path.join(root, "work_st")
    c.freeze()
    tr = TRAINER[c.trainer](c, 0)
    tr.run()


def test_validate_gpu_matches_oracle_miou(world):
    from oracle import deeplab_ref, metrics_ref
    from hiast_amd.workflows.validator import Validator
    cfg, sd, root = world
    c = cfg.clone()
    c.model.type = "SourceOnlySegmentor"
    c.freeze()
    v = Validator(c, device=torch.device("cuda"))
    miou = v.run()
    inter = np.zeros(C, np.int64)
    union = np.zeros(C, np.int64)
    torch.set_num_threads(16)
    from hiast_amd.sseg.datasets import utils as du
    for data in v.v_loader:
        imgs = data["images"]
        if imgs.dtype == torch.uint8:      # the validator normalises on the device (round 6); the oracle takes the HOST transform
            imgs = torch.stack([du._img_to_tensor(i.numpy(), du.MEAN, du.STD) for i in imgs])
        with torch.no_grad():
            logits, _, _ = deeplab_ref.segmentor_logits(imgs, sd)
        pred = torch.softmax(logits, 1).argmax(1).numpy()
        i, u = metrics_ref.intersection_and_union(pred, data["labels"].numpy().astype(np.int64), C)
        inter += i
        union += u
    want, _, _ = metrics_ref.miou(inter, union)
    assert abs(miou - want) <= 0.05 / 100 + 1e-9, (miou, want)       # north_star: mIoU equal to reference +-0.05 (points)


def _write_yaml(cfg, path, extra=None):
    import yaml
    d = cfg.to_dict()
    if extra:
        for k, v in extra.items():
            node = d
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = v
    with open(path, "w") as f:
        yaml.safe_dump(d, f)


def test_cli_entry_points(world, monkeypatch):
    """the three reference CLIs (same flags) through `python -m hiast_amd.<script>`'s main(): generate -> train
    -> validate, configured by YAML files exactly like the reference's configs/*.yaml"""
    import importlib
    import hiast_amd.utils.default_config as dc
    cfg, sd, root = world
    ydir = os.path.join(root, "yaml")
    os.makedirs(ydir, exist_ok=True)
    base = os.path.join(ydir, "sl.yaml")
    _write_yaml(cfg, base, {"pseudo_policy.save_dir": None, "train.total_iter": 2, "train.iter_report": 1,
                            "train.iter_val": 100, "train.batch_size": 2,
                            "dataset.target.aug_type": ["PRS-%d-%d" % (H, W)], "trainer": "SelfTrainingTrainer"})

    def fresh(modname):
        # every reference script mutates the module-global cfg: give each CLI call a fresh default tree
        dc.cfg = dc.get_default_cfg()
        m = importlib.import_module(modname)
        return importlib.reload(m)

    save_dir = os.path.join(root, "cli_pseudo", "pseudo_labels")
    gen = fresh("hiast_amd.generate_pseudo_labels")
    gen.main(["--config_file", base, "--pseudo_resume_from", cfg.pseudo_policy.resume_from,
              "--pseudo_save_dir", save_dir, "--batch_size", "3"])
    assert len(os.listdir(save_dir)) == 6
    assert os.path.exists(os.path.join(save_dir, "..", "samples_with_class.json"))

    monkeypatch.setenv("WORLD_SIZE", "1")
    tr = fresh("hiast_amd.train")
    work = os.path.join(root, "cli_work")
    tr.main(["--config_file", base, "--resume_from", cfg.train.resume_from, "--pseudo_save_dir", save_dir,
             "--work_dir", work])
    assert os.path.exists(os.path.join(work, "sl.yaml")) and os.path.exists(os.path.join(work, "train.log"))

    val = fresh("hiast_amd.validate")
    vy = os.path.join(ydir, "validate.yaml")
    _write_yaml(cfg, vy, {"model.type": "SourceOnlySegmentor", "pseudo_policy.save_dir": None})
    miou = val.main(["--config_file", vy, "--resume_from", cfg.validate.resume_from, "--device", "cuda"])
    assert 0.0 <= miou <= 1.0


def test_training_resumes_after_validation(world):
    """iter_val < total_iter: the validation pass switches the WRAPPER to eval() and back (base_trainer.py:161,
    the reference calls model.train() at the top of every iteration); the iterations after it must run in train mode
    (batch statistics, losses dict) — they raised NotImplementedError when only the inner module was switched."""
    from hiast_amd.utils.registry.registries import TRAINER
    cfg, sd, root = world
    c = cfg.clone()
    c.trainer = "SelfTrainingTrainer"
    c.dataset.target.pseudo_dir = cfg.pseudo_policy.save_dir
    c.dataset.target.aug_type = ["PRS-%d-%d" % (H, W)]
    c.train.gpu_num = 1
    c.train.batch_size = 2
    c.train.total_iter = 3
    c.train.iter_report = 1
    c.train.iter_val = 1
    c.work_dir = os.path.join(root, "work_st_val")
    c.freeze()
    tr = TRAINER[c.trainer](c, 0)
    tr.run()
    assert tr.model.training and tr.model.module.training
    rm = tr.model.module.seg_model.backbone.bn1.num_batches_tracked
    assert int(rm) == 3, "every iteration must have run BatchNorm in train mode"


def test_config5_synthia_source_round(tmp_path):
    """BASELINE configs[4] in miniature: SYNTHIA -> Cityscapes.  A 19-class model, classes {9, 14, 16} absent from the
    data (SYNTHIA has no terrain / truck / train): IAS pseudo labels -> HIAST training iterations with CopyPaste on
    the SYNTHIA class values (absent classes never drawn; the reference's NaN probabilities documented in DESIGN §8)
    -> validation with the 16- and 13-class rescale (workflows/validator.py:108-113) equal to the oracle's."""
    from oracle import cref, deeplab_ref, metrics_ref
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import MODEL, PSEUDO_POLICY, TRAINER
    from hiast_amd.tools import synth_data
    from hiast_amd.workflows.validator import Validator
    from make_golden import seeded_state_dict
    root = str(tmp_path)
    cfg = synth_data.synthetic_cfg(root, n_train=6, n_val=4, h=H, w=W, source_type="SYNTHIA")
    m = MODEL["SelfTrainingSegmentor"](cfg)
    sd = {"seg_model." + k: v for k, v in seeded_state_dict(m.seg_model, 781).items()}
    for i in range(4):
        sd["seg_model.aspp.conv2d_list.%d.weight" % i] = sd["seg_model.aspp.conv2d_list.%d.weight" % i] * 25.0
        # a net trained on SYNTHIA never predicts the three absent classes
        sd["seg_model.aspp.conv2d_list.%d.weight" % i][[9, 14, 16]] = 0.0
        sd["seg_model.aspp.conv2d_list.%d.bias" % i][[9, 14, 16]] = -250.0
    ck = os.path.join(root, "synthia_warmup.pth")
    torch.save(sd, ck)
    cfg.pseudo_policy.resume_from = cfg.train.resume_from = cfg.validate.resume_from = ck
    gen = PSEUDO_POLICY["IAS"](cfg)
    gen.run()
    means = np.load(os.path.join(cfg.pseudo_policy.save_dir, "..", "class_mean_probabilities.npy"))
    stats = np.load(os.path.join(cfg.pseudo_policy.save_dir, "..", "statics_class.npy"))
    assert stats[[9, 14, 16]].sum() == 0 and (means[[9, 14, 16]] == 0).all() and stats.sum() > 0

    c = cfg.clone()
    c.trainer = "ConsistencySelfTrainingTrainer"
    c.dataset.target.pseudo_dir = cfg.pseudo_policy.save_dir
    c.dataset.target.aug_type = ["PRS-%d-%d" % (H, W), "CCA"]
    c.cst_training.is_enabled = True
    c.cst_training.cst_loss.weight = 0.5
    c.preprocessor.type = "CopyPaste"
    c.train.gpu_num, c.train.batch_size, c.train.total_iter, c.train.iter_report, c.train.iter_val = 1, 2, 2, 1, 2
    c.work_dir = os.path.join(root, "work_synthia")
    c.freeze()
    tr = TRAINER[c.trainer](c, 0)
    cp = tr.preprocessor
    assert cp.ignored_classes == [9, 14, 16] and not set(cp.hard_classes.tolist()) & {9, 14, 16}
    assert np.isfinite(cp.class_probs).all() and (cp.class_probs[[9, 14, 16]] == 0).all()
    tr.run()

    v = Validator(cfg, device=torch.device("cuda", 0))
    miou16 = v.run()
    inter, union = np.zeros(C, np.int64), np.zeros(C, np.int64)
    torch.set_num_threads(16)
    from hiast_amd.sseg.datasets import utils as du
    for data in v.v_loader:
        imgs = data["images"]
        if imgs.dtype == torch.uint8:      # the validator normalises on the device (round 6); the oracle takes the HOST transform
            imgs = torch.stack([du._img_to_tensor(i.numpy(), du.MEAN, du.STD) for i in imgs])
        with torch.no_grad():
            z = deeplab_ref.segmentor_logits(imgs, sd)[1].numpy()
        _, lab = cref.tta([z], None, [(H, W)], H, W, want_probs=False)
        a, b = metrics_ref.intersection_and_union(lab.astype(np.int64), data["labels"].numpy().astype(np.int64), C)
        inter += a
        union += b
    w16, w13, iou = metrics_ref.miou(inter.astype(np.float64), union.astype(np.float64), synthia=True)
    assert abs(100 * miou16 - 100 * w16) <= 0.05 and abs(100 * v.miou_13 - 100 * w13) <= 0.05, (miou16, w16, v.miou_13, w13)
    assert iou[9] == 0 and iou[14] == 0 and iou[16] == 0 and w16 > 0


def test_bench_pseudo_label_pass_on_its_own_stream_gives_the_same_labels(monkeypatch):
    """bench.py enqueues the pseudo-label forward on a stream of its own beside the training forwards; the label map, the
    thresholds and the per-class statistics equal those of the in-order pass (same kernels, same batch, same weights)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod_e2e", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    torch.cuda.set_device(0)
    cfg = bench.make_cfg(1, "ConsistencySelfTrainingTrainer")
    hp = bench.HotPath(cfg, torch.device("cuda", 0), 0, 1, 8)
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("HIAST_BENCH_PL_STREAM", flag)
        hp.thr = 0.9 * np.ones(19)
        mp, am = hp.plabel_begin_async()
        out, teacher = hp.train_forward()                  # the forwards that run beside it in a step
        plbl = hp.plabel_finish(mp, am)
        torch.cuda.synchronize()
        res[flag] = (plbl.cpu().numpy().copy(), hp.thr.copy(), [t.cpu().numpy().copy() for t in hp._stats])
        assert hp._pl_join == (flag == "1")
    assert np.array_equal(res["0"][0], res["1"][0])
    assert np.array_equal(res["0"][1].view(np.uint64), res["1"][1].view(np.uint64))
    for a, b in zip(res["0"][2], res["1"][2]):
        assert np.array_equal(a, b)


def test_bench_steps_run_on_finite_data():
    """bench.py's synthetic state must stay finite (a plain random init drives the IAS thresholds above every pixel within
    two steps: empty confident set, 0/0 losses, NaN weights from the third step on — and the chip clocks ~12 % higher on
    that constant data than on real tensors, which rounds 1-2 measured without noticing): six steps, every loss finite,
    a confident share between 5 % and 95 %"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod_fin", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    torch.cuda.set_device(0)
    cfg = bench.make_cfg(1, "ConsistencySelfTrainingTrainer")
    hp = bench.HotPath(cfg, torch.device("cuda", 0), 0, 1, 8)
    for i in range(6):
        losses, plbl = hp.step()
        vals = {k: float(torch.mean(v)) for k, v in losses.items()}
        assert all(np.isfinite(v) for v in vals.values()), (i, vals)
        share = float((plbl != 255).float().mean())
        assert 0.05 < share < 0.95, (i, share)
    for p in hp.model.parameters():
        assert bool(torch.isfinite(p).all())


def _sharded_gen_worker(rank, world_size, port, cfg_dict, save_dir, batch, backend="gloo", split="2"):
    import sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    if world_size > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":                        # RCCL: one rank per device
            torch.cuda.set_device(rank)
        dist.init_process_group(backend, rank=rank, world_size=world_size)
    else:
        os.environ["HIAST_EVAL_SPLIT"] = split      # batches of 4 forwarded as sub-batches, like the ranks' local batches
    if not (world_size > 1 and backend == "nccl"):
        torch.cuda.set_device(0)
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import PSEUDO_POLICY
    from hiast_amd.utils.default_config import CfgNode
    c = CfgNode(cfg_dict)
    c.pseudo_policy.batch_size = batch
    c.pseudo_policy.save_dir = save_dir
    gen = PSEUDO_POLICY["IAS"](c)
    gen.run()
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_sharded_generation_on_hip_equals_single_process(world, tmp_path, backend):
    """configs[1] sharded over 2 ranks (gloo, both on cuda:0) with the HIP engine — the pipelined loop with its histogram /
    class-sum exchanges on the second stream — writes what ONE process writes at batch = 2 x the local batch: thresholds
    (float64 bit patterns), statistics, label maps.  (8 images; the single process forwards its batches of 4 as two
    sub-batches of 2 and runs in a fresh process like the ranks: the library's stem convolution picks its algorithm by
    batch size and by what the process has run before, everything else is per image.)"""
    import socket
    import torch.multiprocessing as mp
    from PIL import Image
    from hiast_amd.tools import synth_data
    if backend == "nccl" and torch.cuda.device_count() < 2:      # collected everywhere, runs once >= 2 devices are visible
        pytest.skip("RCCL needs one device per rank; %d visible" % torch.cuda.device_count())
    cfg0, sd, _ = world
    root = str(tmp_path)
    cfg = synth_data.synthetic_cfg(root, n_train=8, n_val=1, h=H, w=W)
    cfg.pseudo_policy.resume_from = cfg0.pseudo_policy.resume_from
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    d2 = os.path.join(root, "pseudo_shard2", "pseudo_labels")
    mp.spawn(_sharded_gen_worker, args=(2, port, cfg.to_dict(), d2, 2, backend), nprocs=2, join=True)
    d1 = os.path.join(root, "pseudo_shard1", "pseudo_labels")
    mp.spawn(_sharded_gen_worker, args=(1, port, cfg.to_dict(), d1, 4), nprocs=1, join=True)
    for f in ("class_threshold.npy", "statics_class.npy", "class_mean_probabilities.npy"):
        a, b = np.load(os.path.join(d1, "..", f)), np.load(os.path.join(d2, "..", f))
        assert np.array_equal(a.view(np.uint64) if a.dtype == np.float64 else a,
                              b.view(np.uint64) if b.dtype == np.float64 else b), f
    names = sorted(os.listdir(d1))
    assert names == sorted(os.listdir(d2)) and len(names) == 8
    for n in names:
        assert np.array_equal(np.array(Image.open(os.path.join(d1, n))), np.array(Image.open(os.path.join(d2, n)))), n


def _rehearsal_gen_worker(rank, port, cfg_dict, save_dir, policy, rehearse):
    import sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    torch.cuda.set_device(0)
    if rehearse:
        os.environ["HIAST_DIST_REHEARSAL"] = "1"
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        from hiast_amd.utils import comm
        comm.init_process_group("nccl", rank=0, world_size=1)
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import PSEUDO_POLICY
    from hiast_amd.utils.default_config import CfgNode
    c = CfgNode(cfg_dict)
    c.pseudo_policy.type = policy
    c.pseudo_policy.batch_size = 2
    c.pseudo_policy.save_dir = save_dir
    gen = PSEUDO_POLICY[policy](c)
    assert gen.multi == bool(rehearse)
    gen.run()
    if rehearse:
        from hiast_amd.utils import comm
        assert comm.COUNTS["aux"] > 0                # the exchanges were issued (on RCCL's auxiliary communicator)
        dist.barrier(device_ids=[0])
        dist.destroy_process_group()


@pytest.mark.parametrize("policy", ["IAS", "CBST"])
def test_generator_rehearsal_over_rccl_on_one_rank_equals_single_process(world, tmp_path, policy):
    """the sharded generators' exchange path — histogram / class-sum all-reduce on the auxiliary communicator, CBST's all_gather
    of class counts, `broadcast_object_list` of the resume decision, `all_gather_object` of the per-image records, barriers —
    through torch's RCCL backend on ONE rank (HIAST_DIST_REHEARSAL=1, utils/comm.py): every artefact byte-equal to the plain
    single process.  (RCCL refuses two ranks on one device; the 2- and 4-rank forms of this test run on gloo.)"""
    import socket
    import torch.multiprocessing as mp
    from PIL import Image
    from hiast_amd.tools import synth_data
    cfg0, sd, _ = world
    root = str(tmp_path)
    cfg = synth_data.synthetic_cfg(root, n_train=6, n_val=1, h=H, w=W)
    cfg.pseudo_policy.resume_from = cfg0.pseudo_policy.resume_from
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    d2 = os.path.join(root, "pseudo_rehearsal", "pseudo_labels")
    mp.spawn(_rehearsal_gen_worker, args=(port, cfg.to_dict(), d2, policy, True), nprocs=1, join=True)
    d1 = os.path.join(root, "pseudo_single", "pseudo_labels")
    mp.spawn(_rehearsal_gen_worker, args=(port, cfg.to_dict(), d1, policy, False), nprocs=1, join=True)
    for f in ("class_threshold.npy", "statics_class.npy", "class_mean_probabilities.npy"):
        a, b = np.load(os.path.join(d1, "..", f)), np.load(os.path.join(d2, "..", f))
        assert np.array_equal(a.view(np.uint64) if a.dtype == np.float64 else a,
                              b.view(np.uint64) if b.dtype == np.float64 else b), f
    for f in ("sample_class_stats.json", "samples_with_class.json"):
        assert open(os.path.join(d1, "..", f)).read() == open(os.path.join(d2, "..", f)).read(), f
    names = sorted(os.listdir(d1))
    assert names == sorted(os.listdir(d2)) and len(names) == 6
    for n in names:
        assert np.array_equal(np.array(Image.open(os.path.join(d1, n))), np.array(Image.open(os.path.join(d2, n)))), n


def _rehearsal_round_worker(rank, port, cfg_dict, out):
    import sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["HIAST_DIST_REHEARSAL"] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import TRAINER
    from hiast_amd.utils.default_config import CfgNode
    from hiast_amd.utils import comm
    c = CfgNode(cfg_dict)
    c.train.port = int(port)
    c.freeze()
    tr = TRAINER[c.trainer](c, 0)           # BaseTrainer.initialize starts the one-rank 'nccl' group itself
    ok = dist.is_initialized() and dist.get_backend() == "nccl" and tr.multi
    ok = ok and isinstance(tr.model, torch.nn.parallel.DistributedDataParallel)
    tr.run()
    fin = all(bool(torch.isfinite(p).all()) for p in tr.model.parameters())
    json.dump({"ok": bool(ok), "finite": fin, "stat": comm.COUNTS["stat"], "aux": comm.COUNTS["aux"],
               "tracked": int(tr.model.module.seg_model.backbone.bn1.num_batches_tracked)}, open(out, "w"))
    dist.barrier(device_ids=[0])
    dist.destroy_process_group()


def test_config3_round_rehearsed_over_rccl_on_one_rank(world, tmp_path):
    """the config-3 round of `test_config3_hiast_training_round` (CopyPaste + HIAST trainer, report every iteration, validation +
    checkpoints at the end) with the trainer's N > 1 machinery on torch's RCCL backend, ONE rank (HIAST_DIST_REHEARSAL=1): DDP,
    104 SyncBN layers exchanging on the statistics communicator, the recorder's packed loss all-reduce, the validation's
    intersection / union all-reduce on the auxiliary communicator, rank-0 checkpoints with the reference's state-dict keys"""
    import socket
    import torch.multiprocessing as mp
    cfg, sd, root = world
    c = cfg.clone()
    c.trainer = "ConsistencySelfTrainingTrainer"
    c.dataset.target.pseudo_dir = cfg.pseudo_policy.save_dir
    c.dataset.target.aug_type = ["PRS-%d-%d" % (H, W), "CCA"]
    c.cst_training.is_enabled = True
    c.cst_training.cst_loss.weight = 0.5
    c.preprocessor.type = "CopyPaste"
    c.train.gpu_num, c.train.batch_size, c.train.total_iter, c.train.iter_report, c.train.iter_val = 1, 2, 2, 1, 2
    c.train.lr = 3e-6
    c.work_dir = str(tmp_path / "work_rehearsal")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "r.json")
    mp.spawn(_rehearsal_round_worker, args=(port, c.to_dict(), out), nprocs=1, join=True)
    r = json.load(open(out))
    assert r["ok"] and r["finite"] and r["tracked"] == 2, r
    assert r["stat"] == 2 * 208 and r["aux"] >= 1, r           # two training iterations; the validation areas
    ck = os.path.join(c.work_dir, "checkpoints")
    assert {"model_last.pth", "ema_model_last.pth"} <= set(os.listdir(ck))
    saved = torch.load(os.path.join(ck, "model_last.pth"), map_location="cpu")
    assert list(saved.keys()) == list(sd.keys())          # reference state-dict keys: no 'module.' prefix, SyncBN = BN keys
    assert "mIoU" in open(os.path.join(c.work_dir, "train.log")).read()


def test_sharded_generation_four_ranks_equals_single_process(world, tmp_path):
    """the same at FOUR ranks (gloo, all on cuda:0; a GPU box allows six processes on its card): one image per rank and
    global batch — the reference-semantics split of cfg4 / cfg5 — against one process at batch 4 forwarded as four
    sub-batches of 1: thresholds (float64 bit patterns), statistics and label maps byte for byte"""
    import socket
    import torch.multiprocessing as mp
    from PIL import Image
    from hiast_amd.tools import synth_data
    cfg0, sd, _ = world
    root = str(tmp_path)
    cfg = synth_data.synthetic_cfg(root, n_train=8, n_val=1, h=H, w=W)
    cfg.pseudo_policy.resume_from = cfg0.pseudo_policy.resume_from
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    d4 = os.path.join(root, "pseudo_shard4", "pseudo_labels")
    mp.spawn(_sharded_gen_worker, args=(4, port, cfg.to_dict(), d4, 1, "gloo"), nprocs=4, join=True)
    d1 = os.path.join(root, "pseudo_shard1", "pseudo_labels")
    mp.spawn(_sharded_gen_worker, args=(1, port, cfg.to_dict(), d1, 4, "gloo", "4"), nprocs=1, join=True)
    for f in ("class_threshold.npy", "statics_class.npy", "class_mean_probabilities.npy"):
        a, b = np.load(os.path.join(d1, "..", f)), np.load(os.path.join(d4, "..", f))
        assert np.array_equal(a.view(np.uint64) if a.dtype == np.float64 else a,
                              b.view(np.uint64) if b.dtype == np.float64 else b), f
    names = sorted(os.listdir(d1))
    assert names == sorted(os.listdir(d4)) and len(names) == 8
    for n in names:
        assert np.array_equal(np.array(Image.open(os.path.join(d1, n))), np.array(Image.open(os.path.join(d4, n)))), n
