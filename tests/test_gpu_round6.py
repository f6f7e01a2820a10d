"""Round-6 additions of the boundary (ABI 6): the statistics-row check of the tile-kernel launches, the thread-local
co-scheduling hint, the CU count through the ABI, the CU reserve of the N > 1 path (persistent grids sized to CUs - n, streams
whose kernels cannot be placed on the reserved CUs) and the watchdog of bench.py.  Reference side: the conv -> bn chains of
Bottleneck.forward (code/sseg/models/modules/resnet.py:78-98) under DDP + SyncBN (code/workflows/trainer/base_trainer.py:43-56)."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import synth
from test_gpu_kernels import dev

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def K():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from hiast_amd import kernels
    return kernels


def test_device_cus_comes_from_the_device(K):
    """hiast_device_cus (hipDeviceGetAttribute) == what torch reports for the device; MI355X: 256"""
    n = K.device_cus()
    assert n == torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    assert n >= 64 and K.reserve_cus() == 0 and K.grid_cus() == n


def test_stats_launch_refuses_a_row_count_of_another_tile_form(K):
    """the row count of `partial` and the tile form came from two independent reads of the co-scheduling hint; a
    toggle in between sized the buffer for 256-row tiles and ran 128-row tiles (twice the rows: an overrun).  ABI 6: the
    launch takes the caller's row count and returns HIAST_E_ARG when the form it takes writes another number of rows."""
    from hiast_amd import _lib
    lib = _lib.load()
    B, H, W, Cin, N = 4, 64, 128, 1024, 256             # 128 tiles of 256 rows: half of the chip
    M = B * H * W
    x = dev(synth.normal_f32(6000, (B, H, W, Cin))).half()
    wp = K.pack_conv_weight(dev(synth.normal_f32(6001, (N, Cin, 1, 1), 0.03)), K.FMT_FP16)
    y = torch.empty((B, H, W, N), dtype=torch.float16, device=x.device)
    null = ctypes.c_void_p(0)

    def launch(rows):
        partial = torch.full((M // 128 + 8, N, 2), float("nan"), dtype=torch.float32, device=x.device)   # (room for either form)
        rc = lib.hiast_igemm_bn_act(K._ptr(x), K._ptr(wp), null, null, null, null, 0.0, null, 0, K._ptr(y), B, H, W, Cin, N, 1, 1, 1,
                                    K.FMT_FP16, 0, K._ptr(partial), rows, null, 0, K._stream())
        torch.cuda.synchronize()
        return rc, partial

    alone, cos = M // 128, M // 256
    assert lib.hiast_igemm_stats_rows(M, Cin, N, 1, K.FMT_FP16) == alone         # a launch that runs alone: 128 x 128 tiles
    rc, p = launch(alone)
    assert rc == 0 and bool(torch.isfinite(p[:alone]).all()) and bool(torch.isnan(p[alone:]).all())
    rc, p = launch(cos)                                                         # rows of the OTHER form: refused, nothing written
    assert rc == -1 and bool(torch.isnan(p).all())
    with K.cosched():                                                           # the hint set: the 256-row form, and its rows
        assert lib.hiast_igemm_stats_rows(M, Cin, N, 1, K.FMT_FP16) == cos
        rc, p = launch(cos)
        assert rc == 0 and bool(torch.isfinite(p[:cos]).all()) and bool(torch.isnan(p[cos:]).all())
        assert launch(alone)[0] == -1
        y1, part1 = K.igemm_bn_act(x, wp, 1, None, None, False, want_stats=True)
        assert part1.shape[0] == cos
    y0, part0 = K.igemm_bn_act(x, wp, 1, None, None, False, want_stats=True)    # the hint is gone with the context
    assert part0.shape[0] == alone and torch.equal(y0, y1)
    s0, s1 = K.bn_nhwc_stats_from_partial(part0).double(), K.bn_nhwc_stats_from_partial(part1).double()
    assert float((s0 - s1).abs().max()) <= 1e-5 * max(1.0, float(s0.abs().max()))
    # the data-gradient launch with the BatchNorm-backward sums takes its row count the same way
    bx = dev(synth.normal_f32(6002, (B, H, W, N))).half()
    sm, si = dev(0.1 * synth.normal_f32(6003, (N,))), dev(np.abs(synth.normal_f32(6004, (N,))) + 0.5)
    da = torch.empty_like(bx)
    partial = torch.empty((alone, N, 2), dtype=torch.float32, device=x.device)
    args = (K._ptr(x), K._ptr(wp), K._ptr(da), B, H, W, Cin, N, 1, 1, K._ptr(bx), null, null, K._ptr(sm), K._ptr(si), K._ptr(partial))
    assert lib.hiast_igemm_dgrad_bn_stats(*args, alone, K.FMT_FP16, K._stream()) == 0
    assert lib.hiast_igemm_dgrad_bn_stats(*args, cos, K.FMT_FP16, K._stream()) == -1
    torch.cuda.synchronize()


def test_eval_forward_split_sets_no_environment_variable(K, monkeypatch):
    """the sub-batch forwards pass the co-scheduling hint through the library's thread-local setter: os.environ is not touched
    (setenv / unsetenv around every forward raced with getenv in the DataLoader's and the runtime's threads)"""
    from hiast_amd import functional as HF
    from hiast_amd.utils.default_config import get_default_cfg
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import MODEL
    from make_golden import seeded_state_dict
    m = MODEL["SelfTrainingSegmentor"](get_default_cfg())
    m.load_state_dict({"seg_model." + k: v for k, v in seeded_state_dict(m.seg_model, 9100).items()})
    m = m.cuda().eval()
    x = torch.from_numpy(synth.normal_f32(6100, (2, 3, 128, 256))).cuda()
    seen = []
    orig = K.igemm_bn_act
    lib = K._lib.load()

    def spy(*a, **k):
        prev = lib.hiast_igemm_set_cosched(1)         # read the calling thread's hint (and put it back)
        lib.hiast_igemm_set_cosched(prev)
        seen.append(prev)
        return orig(*a, **k)
    monkeypatch.setattr(K, "igemm_bn_act", spy)
    env_before = dict(os.environ)
    with torch.no_grad():
        one = m(x, lowres=True)["logits_lowres"]
        assert seen and set(seen) == {-1}
        del seen[:]
        two = HF.eval_forward_split(m, x, parts=2)["logits_lowres"]
    assert seen and set(seen) == {1}                    # every launch of the sub-batches ran with the hint set
    assert lib.hiast_igemm_set_cosched(-1) == -1        # ... and it is gone afterwards
    assert dict(os.environ) == env_before
    assert torch.equal(one, two)                        # (an eval forward treats every image alone)


def _run_kernels(K, tag):
    """a small set of launches of every kernel family whose grid or work split follows hiast_grid_cus"""
    out = {}
    B, H, W = 8, 64, 128
    x = dev(synth.normal_f32(6200, (B, H, W, 256))).half()
    w3 = dev(synth.normal_f32(6201, (1024, 256, 1, 1), 0.06))
    wp3 = K.pack_conv_weight(w3, K.FMT_FP16)
    out["xconv_stats_y"], part = K.igemm_bn_act(x, wp3, 1, None, None, False, want_stats=True)      # K9e xconv + statistics
    out["xconv_stats"] = K.bn_nhwc_stats_from_partial(part)
    out["rows"] = K._lib.load().hiast_igemm_stats_rows(B * H * W, 256, 1024, 1, K.FMT_FP16)     # (what the row form would write)
    bn = torch.nn.BatchNorm2d(1024).cuda().eval()
    with torch.no_grad():
        bn.running_mean.copy_(dev(0.1 * synth.normal_f32(6202, (1024,))))
        bn.running_var.copy_(dev(np.abs(synth.normal_f32(6203, (1024,))) + 0.5))
    res = dev(synth.normal_f32(6204, (B, H, W, 1024))).half()
    out["xconv_bn_res_relu"] = K.igemm_bn_act(x, wp3, 1, bn, res, True)
    xp = K.split_planes(x.float().view(-1, 256)).view(B, H, W, 512)
    rp = K.split_planes(res.float().view(-1, 1024)).view(B, H, W, 2048)
    out["xconv2"] = K.igemm_bn_act(xp, K.pack_conv_weight(w3, 2), 2, bn, rp, True)              # K9g
    dy = dev(synth.normal_f32(6205, (B, H, W, 1024))).half()
    d2 = dev(synth.normal_f32(6206, (B, H, W, 256))).half()
    x1 = dev(synth.normal_f32(6207, (B, H, W, 1024))).half()
    dws = K.conv_wgrad_group([(dy, x, 1, 1, 1), (d2, x, 3, 1, 2), (d2, x1, 1, 1, 1)])              # K9d grouped
    for i, t in enumerate(dws):
        out["wgrad_group_%d" % i] = t
    img = dev(synth.normal_f32(6208, (2, 3, 256, 512)))
    wst = dev(synth.normal_f32(6209, (64, 3, 7, 7), 0.05))
    y, p = K.stem_train_fwd(img, wst, K.FMT_FP16)                                                  # K9k
    out["stem_train_y"], out["stem_train_stats"] = y, K.bn_nhwc_stats_from_partial(p)
    z = dev(synth.smooth_logits_lr(6210, 4, 19, 64, 128, 6.0))
    mp, am, hist = K.plabel_pass1(z, 512, 1024)                                                    # K2-K4 persistent
    out["maxprob"], out["argmax"], out["hist"] = mp, am, hist
    torch.cuda.synchronize()
    return out


def test_cu_reserve_sizes_persistent_grids_and_keeps_results(K):
    """hiast_set_reserve_cus(8): the persistent / one-block-per-CU launches plan for 248 CUs (fewer statistics rows, other
    pixel-range splits) — element-wise outputs stay bit-equal (the same products in the same order per element), integer
    outputs bit-equal, sums equal up to their blocking"""
    cus = K.device_cus()
    base = _run_kernels(K, "all CUs")
    prev = K.reserve_cus(5)                  # -> 8: whole rounds of the 8 XCDs
    try:
        assert prev == 0 and K.reserve_cus() == 8 and K.grid_cus() == cus - 8
        capped = _run_kernels(K, "8 reserved")
    finally:
        K.reserve_cus(0)
    assert capped["rows"] < base["rows"] and capped["rows"] % 8 == 0, (base["rows"], capped["rows"])
    for k in ("xconv_stats_y", "xconv_bn_res_relu", "xconv2", "stem_train_y", "maxprob", "argmax", "hist"):
        assert torch.equal(base[k], capped[k]), k
    for k in ("xconv_stats", "stem_train_stats", "wgrad_group_0", "wgrad_group_1", "wgrad_group_2"):
        a, b = base[k].double(), capped[k].double()
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(a.abs().max())), k
    again = _run_kernels(K, "all CUs again")
    assert again["rows"] == base["rows"] and torch.equal(again["wgrad_group_1"], base["wgrad_group_1"])


def test_reserved_stream_runs_the_tile_kernels(K):
    """a stream with a queue CU mask (hiast_stream_create_reserved: 8 CUs — one per XCD — taken out): tile kernels, whose grid is
    their tile count, run on it to the same bits; work on it is ordered with torch's streams through events as usual"""
    B, H, W, C = 4, 64, 128, 256
    x = dev(np.maximum(synth.normal_f32(6300, (B, H, W, C)), 0)).half()
    wp = K.pack_conv_weight(dev(synth.normal_f32(6301, (C, C, 3, 3), 0.03)), K.FMT_FP16)
    want = K.igemm_bn_act(x, wp, 1, None, None, True, 1, 2)
    st = K.reserved_stream(8)
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        got = K.igemm_bn_act(x, wp, 1, None, None, True, 1, 2)
        got2 = K.igemm_bn_act(got, wp, 1, None, None, True, 1, 2)
    torch.cuda.current_stream().wait_stream(st)
    want2 = K.igemm_bn_act(want, wp, 1, None, None, True, 1, 2)
    torch.cuda.synchronize()
    assert torch.equal(got, want) and torch.equal(got2, want2)


def test_bench_watchdog_prints_a_diagnostic_line_and_exits_nonzero():
    """bench.py --watchdog-s: no finished stage for that long -> ONE JSON line with "error" and where the rank stands, exit code 3
    (a thread of the rank itself; nothing is re-executed).  The limit here is shorter than the model build."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--batch", "2", "--watchdog-s", "0.5"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 3, (p.returncode, p.stderr[-2000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["value"] is None and d["error"].startswith("watchdog") and d["n_gpus"] == 1
    wd = d["watchdog"]
    assert wd["rank"] == 0 and wd["limit_s"] == 0.5 and wd["expected_per_step"]["syncbn_stat_all_reduces"] == 208
    assert "[bench watchdog] rank 0" in p.stderr


# ------------------------------------------------------------------------------------------------ generator: shared forwards
@pytest.mark.parametrize("policy", ["IAS", "CT"])
def test_generator_grouped_forwards_write_the_same_artefacts(tmp_path, monkeypatch, policy):
    """round 6: the pipelined generators let consecutive small loader batches share ONE forward (pass 1 and the threshold update
    stay per batch) and keep two such forwards in flight on two streams (HipPlabelEngine.group / .lanes).  Against the round-5
    pipeline — one forward per loader batch, one at a time (HIAST_GEN_GROUP=1, HIAST_GEN_LANES=1) — every artefact is the same
    byte for byte: label maps, thresholds (float64 bit patterns), statistics, class means; batch sizes 1, 2 and 3 over 11 images
    (ragged last group, ragged last batch).  512 x 512 frames: the smallest size that is grouped at all — below it a grouped launch
    would cross the 4096-row gate of the xconv kernels and change summation orders (HipPlabelEngine.group).
    Reference: workflows/pseudo_label_generator.py:181-213, :67-105."""
    from PIL import Image
    from hiast_amd.utils.default_config import get_default_cfg  # noqa: F401
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import MODEL, PSEUDO_POLICY
    from hiast_amd.tools import synth_data
    from make_golden import seeded_state_dict
    Hh, Ww = 512, 512
    root = str(tmp_path)
    cfg = synth_data.synthetic_cfg(root, n_train=11, n_val=1, h=Hh, w=Ww)
    m = MODEL["SelfTrainingSegmentor"](cfg)
    m.load_state_dict({"seg_model." + k: v for k, v in seeded_state_dict(m.seg_model, 777).items()})
    m = m.cuda().eval()
    ds = np.stack([synth_data.make_sample(5 + i, Hh, Ww)[0].astype(np.float32).transpose(2, 0, 1) for i in range(2)]) / 255.0
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

    def run(tag, bs, env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = cfg.clone()
        c.defrost()
        c.pseudo_policy.type = policy
        c.pseudo_policy.resume_from = ck
        c.pseudo_policy.batch_size = bs
        c.pseudo_policy.save_dir = os.path.join(root, "out_%s_%d" % (tag, bs), "pseudo_labels")
        c.dataset.num_workers = 0
        gen = PSEUDO_POLICY[policy](c)
        hw = tuple(c.pseudo_policy.resize_size)
        assert hw == (Hh, Ww)
        if tag == "grouped":
            # the rule counts pixels: 4 bench-size (1024 x 512) images per forward; a 512 x 512 frame is half of one
            assert gen.engine.group(bs, hw) == min(8, -(-8 // bs)) and gen.engine.group(2, (512, 1024)) == 2
            assert gen.engine.group(bs, (128, 256)) == 1 and gen.engine.group(bs, None) == 1      # small / unknown frames: never
            assert gen.engine.group(1, (1024, 2048)) == 1 and gen.engine.lanes(2, (1024, 2048)) == 1
            assert gen.engine.lanes(1, (1024, 2048)) == 2 and gen.engine.lanes(4, (512, 1024)) == 2
        elif tag == "lanes":
            assert gen.engine.group(bs, hw) == 2 and gen.engine.lanes(2 * bs, hw) == 2      # two batches per forward, two lanes
        else:
            assert gen.engine.group(bs, hw) == 1 and gen.engine.lanes(bs, hw) == 1
        gen.run()
        for k in env:
            monkeypatch.delenv(k)
        d = c.pseudo_policy.save_dir
        out = {n: np.array(Image.open(os.path.join(d, n))) for n in sorted(os.listdir(d))}
        for f in ("class_threshold.npy", "statics_class.npy", "class_mean_probabilities.npy"):
            p = os.path.join(d, "..", f)
            if os.path.exists(p):
                a = np.load(p)
                out[f] = a.view(np.uint64) if a.dtype == np.float64 else a
        return out

    for bs in (1, 2, 3):
        ref = run("serial", bs, {"HIAST_GEN_GROUP": "1", "HIAST_GEN_LANES": "1"})
        for tag, env in (("grouped", {}), ("lanes", {"HIAST_GEN_GROUP": "2"})):
            got = run(tag, bs, env)
            assert sorted(ref) == sorted(got) and len([k for k in ref if k.endswith(".png")]) == 11
            for k in ref:
                assert np.array_equal(ref[k], got[k]), (policy, bs, tag, k)


def test_validator_batches_a_view_with_its_mirror_image(tmp_path, monkeypatch):
    """round 6: the TTA validator (reference: workflows/validator.py:34-55) runs a resized view and its horizontal flip in ONE
    forward where that does not change the kernels that run (frames >= 512 x 512).  Per-class IoU and mIoU are bit for bit those of
    two forwards (HIAST_VAL_BATCH_FLIP=0); two scales, batch sizes 1 and 2."""
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import MODEL
    from hiast_amd.tools import synth_data
    from hiast_amd.workflows.validator import Validator
    from make_golden import seeded_state_dict
    Hh, Ww = 512, 512
    root = str(tmp_path)
    cfg = synth_data.synthetic_cfg(root, n_train=1, n_val=4, h=Hh, w=Ww)
    m = MODEL["SelfTrainingSegmentor"](cfg)
    m.load_state_dict({"seg_model." + k: v for k, v in seeded_state_dict(m.seg_model, 778).items()})
    m = m.cuda().eval()
    ds = np.stack([synth_data.make_sample(5 + i, Hh, Ww)[0].astype(np.float32).transpose(2, 0, 1) for i in range(2)]) / 255.0
    synth_data.calibrate_bn(m, torch.from_numpy((ds - 0.45) / 0.225).cuda())
    ck = os.path.join(root, "warmup.pth")
    torch.save({k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, ck)
    del m
    out = {}
    for bs in (1, 2):
        for tag, env in (("two", "0"), ("one", "1")):
            monkeypatch.setenv("HIAST_VAL_BATCH_FLIP", env)
            c = cfg.clone()
            c.validate.resume_from = ck
            c.validate.batch_size = bs
            c.validate.is_flip = True
            c.validate.resize_sizes = [[Hh, Ww], [640, 640]]
            v = Validator(c)
            calls = []
            fwd = v.model.forward
            monkeypatch.setattr(v.model, "forward", lambda *a, **k: (calls.append(a[0].shape[0]), fwd(*a, **k))[1])
            v.run()
            out[(bs, tag)] = (np.asarray(v.iou).copy(), v.miou, list(calls))
        two, one = out[(bs, "two")], out[(bs, "one")]
        assert len(one[2]) * 2 == len(two[2]) and set(one[2]) == {2 * bs} and set(two[2]) == {bs}
        assert np.array_equal(two[0].view(np.uint64), one[0].view(np.uint64)) and two[1] == one[1], (bs, two[1], one[1])
        assert 0.0 <= one[1] <= 1.0
