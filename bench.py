"""bench.py — HIAST self-training hot path on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W
(N>1: either under torch.distributed.run, or bare — it then starts its own N ranks before touching a GPU)

One STEP = one pass of the hot path over one batch of B synthetic 1024x512 (W x H) images per GPU,
resident in HBM before the timed region:
  A. pseudo-label pass  (BASELINE configs[1]): eval forward in fp32 -> fused upsample/softmax/argmax/
     histogram kernel -> (N>1: RCCL all-reduce of the 19x15361 histogram) -> IAS thresholds -> select kernel
     -> (N>1: all-reduce of the class sums);
  B. self-training step (BASELINE configs[2], HIAST setting): EMA-teacher forward (no grad) + student
     forward/backward under fp16 autocast with dynamic loss scaling (the reference trains under apex O1;
     --amp-dtype bf16: plain bf16, no scaling), fused 4-term
     region-adaptive loss on the labels of A, Adam step (N>1: DDP/RCCL gradient all-reduce overlapped with
     backward, SyncBN), EMA update.
value = N * B * K / max-over-ranks(time of K steps)  [images/s].

Prints ONE JSON line (contract in the task statement) with `roofline` (dominant hand-written kernel,
timed live with HIP events on the launch stream) and `cpu_baseline` (the CPU oracle on the host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

H, W, C = 512, 1024, 19
ASPP_FLOP_PER_IMG = 2.0 * (H // 8) * (W // 8) * C * 2048 * 36        # 22.95 GFLOP
PEAK_FP32_MFMA = 157.3                                                # TFLOP/s, MI355X_MICROARCH.md


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=12)
    p.add_argument("--warmup", type=int, default=4)
    p.add_argument("--batch", type=int, default=8, help="images per GPU per step")
    p.add_argument("--global-batch", type=int, default=0,
                   help="reference semantics of cfg4 (code/train.py:52-53: the configured batch size is the GLOBAL one, every "
                        "rank takes batch/N images and SyncBN pools the statistics): images per step over ALL ranks; "
                        "overrides --batch with global/N and reports scaling 'strong'.  0 = weak scaling (--batch per GPU)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-size", type=int, nargs=2, default=[H, W],
                   help="H W of the CPU-baseline sample images (BASELINE.md §3: the full 512x1024, ~20 s of CPU work on 16 "
                        "cores; smaller samples are scaled to 512x1024 by the pixel ratio)")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); 'gloo' only for "
                   "functional tests of the N>1 path on a single GPU")
    p.add_argument("--same-device", action="store_true", help="testing: every rank uses cuda:0")
    p.add_argument("--rehearse-dist", action="store_true",
                   help="with --gpus 1: a ONE-rank process group on --backend takes the whole N > 1 code path (DDP, SyncBN "
                        "exchanges on the statistics communicator, histogram on the auxiliary one; HIAST_DIST_REHEARSAL=1, "
                        "hiast_amd/utils/comm.py).  The only way to run torch's RCCL backend on a one-GPU box; the line it "
                        "prints says 'dp1-rehearsal' and is not a bench result")
    p.add_argument("--cpu-threads", type=int, default=64, help="upper bound; the usable CPUs of the box decide")
    p.add_argument("--cpu-reps", type=int, default=3)
    p.add_argument("--amp-dtype", default=os.environ.get("HIAST_BENCH_AMP", "fp16"), choices=["bf16", "fp16"],
                   help="16-bit type of the mixed-precision training step: fp16 = the reference's apex-O1 arithmetic (dynamic "
                        "loss scaling, handled on the device), bf16 = no loss scaling; both run on the same hand-written kernels")
    p.add_argument("--watchdog-s", type=float, default=float(os.environ.get("HIAST_BENCH_WATCHDOG_S", "300")),
                   help="seconds without progress (a finished stage / step) after which a rank prints a diagnostic JSON line "
                        "(rank 0: on stdout, in place of the result) and exits with code 3 — a hung collective must not burn a "
                        "whole GPU lease and return nothing; 0 = off")
    p.add_argument("--trainer", default="ConsistencySelfTrainingTrainer",
                   choices=["ConsistencySelfTrainingTrainer", "SelfTrainingTrainer"])
    return p.parse_args()


def make_cfg(world, trainer, amp_dtype="fp16"):
    from hiast_amd.utils.default_config import get_default_cfg
    c = get_default_cfg()
    c.trainer = trainer
    c.train.amp_dtype = amp_dtype
    c.model.type = "SelfTrainingSegmentor"
    c.model.predictor.kld_loss.weight = 0.1        # configs/sl_1.yaml
    c.model.predictor.ent_loss.weight = 1.0
    c.pseudo_policy.type = "IAS"
    c.pseudo_policy.ias.alpha, c.pseudo_policy.ias.beta, c.pseudo_policy.ias.gamma = 0.5, 0.9, 8.0
    c.train.lr, c.train.optimizer, c.train.total_iter = 3e-6, "Adam", 8000
    c.train.gpu_num = world
    if trainer == "ConsistencySelfTrainingTrainer":  # configs/hiast_setting.yaml
        c.cst_training.is_enabled = True
        c.cst_training.cst_loss.weight = 0.5
        c.cst_training.cst_loss.region = "ignored"
    return c


class KernelTimer:
    """HIP-event timing of selected C-ABI launches on torch's current stream (the stream the kernels are
    enqueued on), grouped by launch shape."""

    def __init__(self):
        self.groups = {}
        self.on = False

    def wrap(self, module, name, key_fn):
        orig = getattr(module, name)
        timer = self

        def timed(*a, **k):
            if not timer.on:
                return orig(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = orig(*a, **k)
            e1.record()
            timer.groups.setdefault(key_fn(*a, **k), []).append((e0, e1))
            return y
        setattr(module, name, timed)

    def install(self):
        from hiast_amd import kernels as K

        def aspp_key(x, wpack, Cout, dil, workspace=None):
            return ("aspp_fwd", tuple(x.shape), Cout)

        def ig_key(x, wp, planes, bn, res, relu, stride=1, dil=1, out_f32=False, **_kw):
            return ("igemm", tuple(x.shape), tuple(wp.shape), int(planes), int(stride), int(dil), res is not None,
                    bool(out_f32), bn is not None, bool(relu))
        def aspp2_key(x, wt, bias, dil, workspace=None, planes=None):
            return ("aspp2_fwd", tuple(x.shape), int(bias.numel()), 2 if planes == 2 else (0 if x.dtype == torch.float32 else 1),
                    str(x.dtype).replace("torch.", ""))
        def wg_key(jobs):
            return ("wgrad_group",) + tuple((tuple(j[0].shape), int(j[1].shape[3]), int(j[2])) for j in jobs)
        self.wrap(K, "conv_wgrad_group", wg_key)
        self.wrap(K, "aspp_fwd", aspp_key)
        self.wrap(K, "aspp2_fwd", aspp2_key)
        self.wrap(K, "igemm_bn_act", ig_key)
        # the other matrix-core launches of the step (counted in step_fractions; not roofline candidates of their own):
        # ("flops", name, algorithmic flop, issued flop)
        def fl(name, alg, iss=None):
            return ("flops", name, float(alg), float(alg if iss is None else iss))
        self.wrap(K, "igemm_dgrad_bn_stats", lambda dy, wpt, *a, **k: fl(
            "dgrad+bn sums %s" % (tuple(wpt.shape),), 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * wpt.shape[0] * wpt.shape[1] * wpt.shape[2]))
        self.wrap(K, "xconv_dgrad_gated_bn_stats", lambda dy, wpt, *a, **k: fl(
            "xconv_bs %s" % (tuple(wpt.shape),), 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * wpt.shape[0] * wpt.shape[2]))
        wgf = lambda dy, x, kk, *a, **k: fl("wgrad %dx%d %d->%d" % (kk, kk, x.shape[3], dy.shape[3]),
                                            2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * dy.shape[3] * x.shape[3] * kk * kk)
        self.wrap(K, "conv_wgrad_nhwc", wgf)
        self.wrap(K, "conv_wgrad_small_nhwc", wgf)
        self.wrap(K, "igemm_dgrad_s2", lambda dy, wpt, Hh, Ww: fl(      # transposed stride-2 3x3: three of four taps multiply zeros
            "dgrad_s2", 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * 9 * wpt.shape[0] * wpt.shape[2],
            2.0 * dy.shape[0] * Hh * Ww * 9 * wpt.shape[0] * wpt.shape[2]))
        self.wrap(K, "aspp2_bwd", lambda x, dy, wd, dil, want_dx=True, want_dw=True, workspace=None: fl(
            "aspp2_bwd", (int(want_dx) + int(want_dw)) * 2.0 * x.shape[0] * x.shape[2] * x.shape[3] * x.shape[1] * 33 * dy.shape[1],
            (int(want_dx) + int(want_dw)) * 2.0 * x.shape[0] * x.shape[2] * x.shape[3] * x.shape[1] * 640))
        stem_alg = lambda x: 2.0 * x.shape[0] * ((x.shape[2] - 1) // 2 + 1) * ((x.shape[3] - 1) // 2 + 1) * 64 * 147
        self.wrap(K, "stem_eval", lambda x, weight, bn, fmt: fl("stem_eval", stem_alg(x),
                                                                 stem_alg(x) * 224.0 / 147.0 * (3.0 if int(fmt) == 2 else 1.0)))
        self.wrap(K, "stem_train_fwd", lambda x, weight, fmt: fl("stem_train_fwd", stem_alg(x), stem_alg(x) * 224.0 / 147.0))
        self.wrap(K, "stem_wgrad", lambda x, dy: fl("stem_wgrad", stem_alg(x), stem_alg(x) * 224.0 / 147.0))

    def summary(self):
        out = []
        for key, pairs in self.groups.items():
            ms = [a.elapsed_time(b) for a, b in pairs]
            out.append((key, float(np.mean(ms)), len(ms), float(np.sum(ms))))   # (launches are sampled: see main())
        return sorted(out, key=lambda t: -t[3])


def roofline_of(key, avg_ms, n, steps):
    """-> roofline dict for one launch group"""
    if key[0] == "aspp_fwd":
        B, Cin, h, w = key[1]
        flop = 2.0 * h * w * key[2] * Cin * 36 * B
        ach = flop / (avg_ms * 1e-3) / 1e12
        d = {"kernel": "hiast::aspp_fwd16_kernel<4,3>", "bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA,
             "unit": "TFLOP/s", "frac": ach / PEAK_FP32_MFMA, "traffic": None, "avg_launch_ms": avg_ms,
             "launches_per_step": n / steps,
             "note": "exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) peak 157.3; algorithmic %.2f GFLOP per launch (%d images)"
                     % (flop / 1e9, B)}
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_aspp_fwd.json")
        if os.path.exists(pmc) and B == 8:
            pj = json.load(open(pmc))
            d["traffic"] = pj["hbm_bytes_x2_fetch"]
            d["note"] += ("; HBM bytes/launch from %s = (2*FETCH_SIZE + WRITE_SIZE)*1024, uncorrected %.0f MB, "
                          "algorithmic %.0f MB" % (pj["source"], pj["hbm_bytes_uncorrected"] / 1e6,
                                                   pj["algorithmic_bytes"] / 1e6))
        return d
    if key[0] == "wgrad_group":
        # grouped weight gradients of one bottleneck: jobs = ((dy shape [B,H,W,Cout], Cin, k), ...)
        flop = sum(2.0 * j[0][0] * j[0][1] * j[0][2] * j[0][3] * j[1] * j[2] ** 2 for j in key[1:])
        operands = sum(2.0 * j[0][0] * j[0][1] * j[0][2] * (j[0][3] + j[1]) for j in key[1:])
        tiles = sum((j[0][3] // 256) * (j[1] // 256) * j[2] ** 2 for j in key[1:])
        partial = 2.0 * tiles * (256 // tiles) * 256 * 256 * 4        # written by the blocks, read by the reduction
        ach = flop / (avg_ms * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r05_pmc_igemm.json")
        if os.path.exists(pmc) and sorted(tuple(j) for j in key[1:]) == sorted([((8, 64, 128, 1024), 256, 1),
                                                                                 ((8, 64, 128, 256), 256, 3),
                                                                                 ((8, 64, 128, 256), 1024, 1)]):
            for ent in json.load(open(pmc)).get("kernels", []):
                if ent.get("name") == "wgrad_group_l3":
                    traffic = ent["hbm_bytes_per_launch"]       # (the group kernel alone; the reduction re-reads the partials)
        return {"kernel": "hiast::wgrad_group_kernel + wgrad_group_reduce_kernel (the %d weight gradients of a bottleneck in one "
                          "launch: transposed-read GEMM over the pixel index, fixed-order reduce)" % (len(key) - 1),
                "bound": "mfma", "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": ach / 2500.0, "traffic": traffic,
                "avg_launch_ms": avg_ms, "launches_per_step": n / steps,
                "note": "algorithmic %.1f GFLOP per launch pair (%s); operands %.0f MB + fp32 partial tiles %.0f MB (written and "
                        "read once) -> %.2f TB/s = %.0f%% of the 8 TB/s HBM roof; %d tiles x %d pixel ranges"
                        % (flop / 1e9, ", ".join("%dx%d %d->%d" % (j[2], j[2], j[1], j[0][3]) for j in key[1:]),
                           operands / 1e6, partial / 1e6, (operands + partial) / (avg_ms * 1e-3) / 1e12,
                           100.0 * (operands + partial) / 8e12 / (avg_ms * 1e-3), tiles, 256 // tiles)}
    if key[0] == "aspp2_fwd":
        # whole ASPP head forward (tap GEMM + 33-tap shift-add), the kernel group the north star names
        shp, Cout, mode = key[1], key[2], key[3]
        if mode == 2:
            B, h, w, C2 = shp
            Cin = C2 // 2
        else:
            B, Cin, h, w = shp
        flop = 2.0 * h * w * Cout * Cin * 36 * B
        ach = flop / (avg_ms * 1e-3) / 1e12
        peak = 2500.0 / 3.0 if mode != 1 else 2500.0
        alg_bytes = B * h * w * Cin * (2 if mode == 1 else 4) + 33 * Cout * Cin * 4 + B * Cout * h * w * 4
        return {"kernel": "hiast_aspp2_fwd (%s tap GEMM on hiast::igemm_bn_act_kernel + aspp2_shift_add_kernel)"
                          % ("split-bf16" if mode != 1 else {"float16": "fp16", "bfloat16": "bf16"}.get(key[4], key[4])),
                "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                "avg_launch_ms": avg_ms, "launches_per_step": n / steps,
                "note": "ASPP head forward, %d images: algorithmic %.1f GFLOP (36 taps x %d x %d) and %.0f MB (feature "
                        "read ONCE + weights + logits) -> %.2f TB/s = %.0f%% of the 8 TB/s HBM peak: arithmetic intensity "
                        "%.0f FLOP/B puts the head on the MFMA roof, not the HBM roof (SURVEY 7.1); the direct exact-fp32 "
                        "form it replaced took 1.9 ms" % (B, flop / 1e9, Cout, Cin, alg_bytes / 1e6,
                                                           alg_bytes / (avg_ms * 1e-3) / 1e12,
                                                           100.0 * alg_bytes / 8e12 / (avg_ms * 1e-3), flop / alg_bytes)}
    # ("igemm", x [B,H,W,PL*Cin], wp [Cout,taps,PL*Cin], PL, stride, dil, has_res, out_f32)
    B, Hh, Ww, CC = key[1]
    Cout, taps, _ = key[2]
    PL, stride, dil, has_res, out_f32, has_bn, relu = key[3:10]
    Cin = CC // PL
    Ho, Wo = (Hh, Ww) if taps == 1 else ((Hh - 1) // stride + 1, (Ww - 1) // stride + 1)
    M = B * Ho * Wo
    flop = 2.0 * M * taps * Cin * Cout
    ach = flop / (avg_ms * 1e-3) / 1e12
    peak = 2500.0 / 3.0 if PL == 2 else 2500.0
    alg_bytes = (B * Hh * Ww * CC + Cout * taps * CC) * 2 + M * Cout * (4 if out_f32 else 2 * PL) * (2 if has_res else 1)
    traffic = traffic_source = None
    traffic_stale = None
    for pmc in ("r06_pmc_igemm.json", "r05_pmc_igemm.json", "r04_pmc_igemm.json", "r03_pmc_igemm.json", "r02_pmc_igemm.json"):       # (the newest collection that has this launch shape)
        path = os.path.join(ROOT, "profiles", pmc)
        if traffic is None and os.path.exists(path):
            doc = json.load(open(path))
            for ent in doc.get("kernels", []):
                if ent.get("key") == [B, Hh, Ww, Cin, Cout, taps, PL, dil]:
                    traffic = ent["hbm_bytes_per_launch"]
                    # collected on THESE kernel sources?  (files of earlier rounds carry no fingerprint: another build)
                    from hiast_amd import _lib as _L
                    traffic_stale = doc.get("kernel_sources_sha16") != _L.kernel_sources_sha16()
                    # NOT measured in this run: PMC counters need their own rocprofv3 --pmc passes (tools/pmc_igemm.sh)
                    traffic_source = ("profiles/%s (round %s: separate rocprofv3 --pmc passes over this launch shape, "
                                      "(2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch; read from the file, not collected by "
                                      "this bench run%s)" % (pmc, pmc[2], "; STALE: collected on a build of other kernel "
                                                             "sources than the running library" if traffic_stale else
                                                             "; same kernel sources as the running library"))
    is_xconv = (PL == 1 and taps == 1 and Cin == 256 and Cout % 512 == 0 and not out_f32 and M >= 4096
                and os.environ.get("HIAST_XCONV", "1") != "0" and not (has_bn and has_res and not relu))
    is_xconv2 = (PL == 2 and taps == 1 and Cin == 256 and Cout % 256 == 0 and not out_f32 and M >= 4096 and has_bn
                 and os.environ.get("HIAST_XCONV2", "1") != "0" and not (has_res and not relu))
    name = ("hiast::xconv_kernel<256,...> (register-resident weights, persistent over 64-row panels; %s%s%s)"
            % ("BN" if has_bn else "plain: student forward / data gradient", " + residual" if has_res else "",
               " + ReLU" if relu else "")) if is_xconv else \
        ("hiast::xconv2_kernel (split-bf16 planes, register-resident weights, persistent over 32-row panels; BN%s%s)"
         % (" + residual" if has_res else "", " + ReLU" if relu else "")) if is_xconv2 else \
        "hiast::igemm_bn_act_kernel<PL=%d,%s,taps=%d> (%s LDS-DMA implicit GEMM%s%s%s)" % (
        PL, "f32out" if out_f32 else "16-bit out", taps, "split-bf16" if PL == 2 else "bf16",
        " + BN" if has_bn else " (plain: student forward / data gradient)", " + residual" if has_res else "",
        " + ReLU" if relu else "")
    # the binding roof is the one the launch sits closer to: MFMA (algorithmic flops against the dense bf16 peak, /3 for
    # split planes) or HBM (algorithmic bytes against 8 TB/s)
    t = avg_ms * 1e-3
    f_mfma, hbm_tbs = ach / peak, alg_bytes / t / 1e12
    f_hbm = hbm_tbs / 8.0
    flavour = ("fp32-equivalent flops: the kernel issues 3 bf16 MFMA flops per algorithmic flop (hi*hi + lo*hi + hi*lo), "
               "MFMA ceiling = dense bf16 peak 2500/3 TFLOP/s" if PL == 2 else "plain bf16 MFMA, ceiling 2500 TFLOP/s dense")
    shape = "B=%d %dx%d Cin=%d Cout=%d taps=%d dil=%d" % (B, Hh, Ww, Cin, Cout, taps, dil)
    # both conventions, side by side: issued MFMA flops (3 per algorithmic flop on split planes) and algorithmic flops, each
    # against the plain dense 16-bit peak of 2500 TFLOP/s
    fracs = {"frac_issued": ach * (3.0 if PL == 2 else 1.0) / 2500.0, "frac_algorithmic_vs_dense": ach / 2500.0,
             "traffic_source": traffic_source, "traffic_stale": traffic_stale}
    if f_hbm > f_mfma:
        return {"kernel": name, "bound": "hbm", "achieved": hbm_tbs * 1e3, "peak": 8000.0, "unit": "GB/s", "frac": f_hbm,
                **fracs, "traffic": traffic, "avg_launch_ms": avg_ms, "launches_per_step": n / steps,
                "note": "algorithmic %.0f MB per launch (%s: input + weights + output%s, 16-bit%s); the MFMA side: %.1f "
                        "GFLOP -> %.0f TFLOP/s = %.0f%% of its ceiling (%s)"
                        % (alg_bytes / 1e6, shape, " + residual" if has_res else "", " hi|lo pairs" if PL == 2 else "",
                           flop / 1e9, ach, 100.0 * f_mfma, flavour)}
    return {"kernel": name, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": f_mfma,
            **fracs, "traffic": traffic, "avg_launch_ms": avg_ms, "launches_per_step": n / steps,
            "note": "algorithmic %.1f GFLOP per launch (%s); %s; algorithmic HBM bytes per launch %.0f MB -> %.2f TB/s = "
                    "%.0f%% of the 8 TB/s HBM roof" % (flop / 1e9, shape, flavour, alg_bytes / 1e6, hbm_tbs, 100.0 * f_hbm)}


def step_fractions(groups, ms_per_step, serial_step_ms):
    """step-level figures that travel with the line: matrix-pipe and HBM use of the WHOLE step.
    Flops: counted live from the matrix-core launches the first timed step observed (every tile-kernel / xconv / data-gradient /
    weight-gradient / ASPP / stem launch with its shape: 3 issued bf16 products per algorithmic flop on split planes, the zero
    taps of the transposed stride-2 launch and the stem's K padding as issued only) — the exact-fp32 and VALU kernels (losses,
    BatchNorm, pass 1 / 2) carry no MFMA flops.  HBM bytes and the serial kernel
    time need their own profiler passes (tools/pmc_bench.sh, tools/prof_bench.sh): read from profiles/r06_step_totals.json when it
    was collected on these kernel sources, else null."""
    alg = iss = 0.0
    for key, _avg, n, _tot in groups:
        if key[0] == "igemm":
            B, Hh, Ww, CC = key[1]
            Cout, taps, _ = key[2]
            PL, stride = key[3], key[4]
            Ho, Wo = (Hh, Ww) if taps == 1 else ((Hh - 1) // stride + 1, (Ww - 1) // stride + 1)
            f = 2.0 * B * Ho * Wo * taps * (CC // PL) * Cout * n
            alg += f
            iss += f * (3.0 if PL == 2 else 1.0)
        elif key[0] == "wgrad_group":
            f = sum(2.0 * j[0][0] * j[0][1] * j[0][2] * j[0][3] * j[1] * j[2] ** 2 for j in key[1:]) * n
            alg += f
            iss += f
        elif key[0] == "flops":
            alg += key[2] * n
            iss += key[3] * n
        # ("aspp2_fwd": its tap GEMM is an igemm launch and counted there; "aspp_fwd": the exact-fp32 head is not on the step)
    out = {"step_tflop_algorithmic_counted": alg / 1e12, "step_tflop_issued_counted": iss / 1e12,
           "step_frac_mfma_issued": iss / (ms_per_step * 1e-3) / 2.5e15,
           "step_frac_mfma_algorithmic": alg / (ms_per_step * 1e-3) / 2.5e15,
           "serial_step_ms": serial_step_ms,
           "step_frac_hbm": None, "serial_kernel_ms": None, "dispatches_per_step": None, "step_totals_source": None}
    path = os.path.join(ROOT, "profiles", "r06_step_totals.json")
    if os.path.exists(path):
        from hiast_amd import _lib as _L
        doc = json.load(open(path))
        stale = doc.get("kernel_sources_sha16") != _L.kernel_sources_sha16()
        out["step_frac_hbm"] = doc["hbm_gb_per_step"] * 1e9 / (ms_per_step * 1e-3) / 8e12
        out["step_hbm_gb"] = doc["hbm_gb_per_step"]
        out["serial_kernel_ms"] = doc.get("serial_kernel_ms")
        out["dispatches_per_step"] = doc.get("dispatches_per_step")
        out["step_totals_source"] = "profiles/r06_step_totals.json (%s)%s" % (
            doc.get("source", "?"), "; STALE: collected on other kernel sources" if stale else "; same kernel sources")
        out["step_totals_stale"] = stale
    return out


class HotPath:
    def __init__(self, cfg, device, rank, world, B, multi=None):
        from hiast_amd.utils import utils
        from hiast_amd.utils.registry import register  # noqa: F401
        from torch.nn.parallel import DistributedDataParallel as DDP
        from hiast_amd.workflows.trainer.base_trainer import _Bare, autocast_dtype, make_grad_scaler
        self.cfg, self.device, self.rank, self.world, self.B = cfg, device, rank, world, B
        self.multi = multi = (world > 1) if multi is None else multi      # the N > 1 code path (also: the one-rank rehearsal)
        utils.seed_everything(cfg.train.random_seed)
        model = utils.init_model(cfg).to(device)
        if os.environ.get("HIAST_BENCH_RAW_INIT", "0") != "1":      # =1: the plain random init — NaN weights from step 3 on
            self._spread_head(model, B)                             # (rounds 1-2 were measured like that: DESIGN §6)
        self.opt, _ = utils.init_optimizers(cfg, model)
        self.sched = utils.init_schedulers(cfg, self.opt)
        self.amp = autocast_dtype(cfg)
        self.scaler = make_grad_scaler(self.amp)        # fp16: apex's dynamic loss scaling (2^16, halved on overflow)
        self.skipped = 0
        self.model = (DDP(model, device_ids=[device.index], gradient_as_bucket_view=True, bucket_cap_mb=32,
                          broadcast_buffers=False) if multi else _Bare(model))
        self.teacher = cfg.cst_training.is_enabled
        if self.teacher:
            self.ema = utils.init_model(cfg, student_model=self.model).to(device)
            for p in self.ema.parameters():
                p.requires_grad = False
            self.ema_updater = utils.EmaUpdater()
            from hiast_amd import functional as HF
            self.side = HF.new_stream(device)
        self.use_side = True
        from hiast_amd import functional as HF
        HF.enable_wgrad_overlap(True)       # train_step() joins the side stream before the optimiser step
        # synthetic batch, resident on the device (normalised float images: what Dataset.transform emits)
        # TWO seeded batches, alternating step by step (round 5): the histogram-contention-sensitive pass 1 and the clocks are
        # not measured on one data distribution only
        self.batches = []
        for i in range(2):
            g = torch.Generator(device="cpu").manual_seed(1234 + rank + 7919 * i)
            weak = torch.randn(B, 3, H, W, generator=g)
            if i == 1:          # a second distribution: smoother images with another contrast
                weak = torch.nn.functional.avg_pool2d(weak, 3, 1, 1) * 2.2 + 0.1
            weak = weak.to(device)
            self.batches.append((weak, (weak * 1.05 + 0.02).contiguous() if self.teacher else weak))
        self.weak, self.strong = self.batches[0]
        self._step_no = 0
        self.thr = 0.9 * np.ones(C)
        self.class_mean_probs = np.zeros(C)
        self._hist_host = self._hist_ready = None
        self.pipelined = os.environ.get("HIAST_BENCH_SERIAL", "0") != "1"

    @staticmethod
    def _spread_head(model, B):
        """A random-init DeepLab predicts a near-uniform softmax (max-prob 0.06-0.1): the IAS thresholds then rise above
        every pixel within two steps, the confident set is empty, the reference's losses are 0/0 = NaN, and from the third
        step on the whole step would run on NaN weights — constant bit patterns on which the chip holds a higher clock
        than on data (MI355X_MICROARCH.md, DVFS items 1 and 7).  Before the optimiser and the EMA copy exist, the random
        init is therefore brought into the state a trained checkpoint is in: (i) the BatchNorm running statistics are set
        to the batch statistics of the synthetic batch (one training-mode forward with momentum 1: the EMA teacher copies
        these buffers from the student after every step and normalises with them), (ii) the classifier weights are scaled
        so that the low-res logits have standard deviation 6 — max-probs spread over 0.2-1.0 as SURVEY.md §8(c) asks for
        configs 2/3.  The step then keeps running on finite, varied data (tools/dbg/soak_bench_steps.py)."""
        dev = next(model.parameters()).device
        g = torch.Generator(device="cpu").manual_seed(1234)
        x = torch.randn(min(B, 4), 3, H, W, generator=g).to(dev)
        bns = [m for m in model.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
        mom = [m.momentum for m in bns]
        was = model.training
        with torch.no_grad():
            for m in bns:
                m.momentum = 1.0
            model.train()
            model(x, lowres=True)
            for m, v in zip(bns, mom):
                m.momentum = v
            model.eval()
            std = float(model(x, lowres=True)["logits_lowres"].float().std())
            k = 6.0 / max(std, 1e-12)
            head = model.seg_model.aspp
            for m in head.conv2d_list:
                m.weight.mul_(k)
                if m.bias is not None:
                    m.bias.mul_(k)
            torch.autograd.graph.increment_version([p for p in head.parameters()])
        model.train(was)

    def _allreduce(self, t):
        if self.multi:          # histogram / class sums of the pseudo-label pass: auxiliary communicator (utils/comm.py)
            from hiast_amd.utils import comm
            comm.all_reduce(t, "aux")
        return t

    # One step = the pseudo-label pass and the training step on the same batch.  Both are split at the point where the
    # host is needed (the IAS threshold update reads the per-class histogram back): the training step's two forward
    # passes do not depend on the pseudo labels, so they are ENQUEUED before the host waits for the histogram — the
    # device works through them while the host computes the thresholds, and the launch queue never runs dry behind
    # the read-back (it did: 0.8 ms of idle device per step + a host that stayed just-in-time for the rest of the step,
    # profiles/r01_s3_critical_before.txt).  No work is skipped or reordered across steps.
    def _graphed(self, net, amp):
        """the inference forward of `net` (no autograd; under autocast(amp) if given) as HF.GraphedEval"""
        from hiast_amd import functional as HF
        g = self.__dict__.setdefault("_graphs", {})
        key = (id(net), amp)
        if key not in g:
            g[key] = HF.GraphedEval(net, amp)
        return g[key]

    def plabel_begin(self):
        """eval forward (fp32-class) + pass 1 + asynchronous read-back of the histogram"""
        from hiast_amd import kernels as K
        net = self.ema if self.teacher else self.model.module     # train.sh generates with the EMA model
        from hiast_amd.utils import utils
        utils.set_mode(net, False)
        with torch.no_grad():
            from hiast_amd import functional as HF
            # fp32 like the reference generator (HIAST_GRAPH_EVAL=1: replayed from a captured HIP graph, except on the steps
            # that time every launch)
            logits = self._graphed(net, None)(self.weak, parts=None if self.use_side else 1, eager=not self.use_side)
            mp, am, hist = K.plabel_pass1(logits.contiguous(), H, W)
            if self.multi and getattr(self, "_pl_on_side", False):
                # N > 1, pass on its own stream: the histogram all-reduce is ISSUED by plabel_finish(), after the student
                # forward has issued its SyncBN all-reduces — a communicator executes its operations in issue order, and
                # this one waits for the whole pseudo-label forward: issued here it would hold back every SyncBN
                # all-reduce behind it, i.e. the student forward would no longer run beside this pass
                self._hist_dev = hist
            else:
                self._hist_exchange(hist)
        return mp, am

    def _hist_exchange(self, hist):
        """all-reduce of the per-class confidence histogram + asynchronous read-back (on the current stream)"""
        hist = self._allreduce(hist)
        if self._hist_host is None:
            self._hist_host = torch.empty(hist.shape, dtype=hist.dtype, pin_memory=True)
            self._hist_ready = torch.cuda.Event()
        self._hist_host.copy_(hist, non_blocking=True)
        self._hist_ready.record()

    def plabel_begin_async(self):
        """plabel_begin() on a stream of its own, beside the two forwards of the training step (which do not depend on
        it): four launch sequences (two pseudo-label sub-batches, teacher, student) keep the matrix pipes busy through each
        other's memory-bound phases (55.9 -> 54.2 ms/step; HIAST_BENCH_PL_STREAM=0: on the main stream).  No work is
        skipped or moved across steps: plabel_finish() joins the stream before pass 2."""
        self._pl_join = False
        if not (self.pipelined and self.use_side and os.environ.get("HIAST_BENCH_PL_STREAM", "1") != "0"):
            return self.plabel_begin()
        if not hasattr(self, "_pl_stream"):
            from hiast_amd import functional as HF
            self._pl_stream = HF.new_stream(self.device)
        self._pl_stream.wait_stream(torch.cuda.current_stream())
        self._pl_on_side = True
        try:
            with torch.cuda.stream(self._pl_stream):
                mp, am = self.plabel_begin()
        finally:
            self._pl_on_side = False
        self._pl_join = True
        return mp, am

    def plabel_finish(self, mp, am):
        """host: thresholds from the histogram (bit-identical to the reference's list + np.quantile); pass 2"""
        from hiast_amd import kernels as K
        from hiast_amd.workflows import ias_math
        with torch.no_grad():
            if getattr(self, "_hist_dev", None) is not None:        # (N > 1: see plabel_begin)
                with torch.cuda.stream(self._pl_stream):
                    self._hist_exchange(self._hist_dev)
                self._hist_dev = None
            if getattr(self, "_pl_join", False):
                torch.cuda.current_stream().wait_stream(self._pl_stream)
                mp.record_stream(torch.cuda.current_stream()); am.record_stream(torch.cuda.current_stream())
            self._hist_ready.synchronize()
            ias = self.cfg.pseudo_policy.ias
            _, self.thr = ias_math.ias_update(self._hist_host.numpy().view(np.uint32), self.thr, ias.alpha, ias.beta,
                                              ias.gamma)
            thr_up = K.h2d_async(ias_math.roundup_f32(self.thr), self.device)
            plbl, count, sfx = K.plabel_pass2(mp, am, thr_up, C)
            cnt = self._allreduce(count.sum(0))
            sfx = self._allreduce(sfx)
        self._stats = (cnt, sfx)
        return plbl

    def plabel_pass(self):
        return self.plabel_finish(*self.plabel_begin())

    def train_forward(self):
        """EMA-teacher forward (no grad) and student forward of the training step -> (student out, teacher logits)"""
        from hiast_amd.utils import utils
        utils.set_mode(self.model, True)
        teacher_lr = None
        main = torch.cuda.current_stream()
        if self.teacher:
            # EMA-teacher forward on a side stream: its MFMA-bound convolutions co-run with the HBM-bound BatchNorm
            # passes of the student forward (ConsistencySelfTrainingTrainer.train_on does the same)
            utils.set_mode(self.ema, False)
            side = self.side if (self.use_side and os.environ.get("HIAST_NO_SIDE_STREAM", "0") != "1") else main
            side.wait_stream(main)
            with torch.cuda.stream(side):
                teacher_lr = self._graphed(self.ema, self.amp)(self.weak, parts=int(os.environ.get("HIAST_BENCH_TEACHER_PARTS", "1")),
                                                               eager=not self.use_side)
        with torch.autocast("cuda", dtype=self.amp, enabled=self.amp is not None):
            out = self.model(self.strong, lowres=True)
        if self.teacher and side is not main:
            main.wait_stream(side)
            teacher_lr.record_stream(main)
        return out, teacher_lr

    def train_finish(self, plbl, out, teacher_lr):
        """fused 4-term loss, backward, Adam, EMA update, scheduler"""
        from hiast_amd import functional as HF
        losses = self.model.module.compute_loss_lowres(out["logits_lowres"], plbl, out["size"], teacher_lr)
        g_loss = sum(torch.mean(v) for v in losses.values())
        self.opt.zero_grad(set_to_none=True)
        (self.scaler.scale(g_loss) if self.scaler else g_loss).backward()
        HF.wgrad_stream_join()      # weight gradients of the trunk run on a side stream in single-process runs
        if self.scaler:             # FusedAdam unscales / skips on the device: no host read of found_inf
            self.scaler.step(self.opt)
            self.scaler.update()
        else:
            self.opt.step()
        if self.teacher:
            self.ema_updater(self.ema, self.model, self.cfg.cst_training.ema_model.gamma)
        for s in self.sched:
            s.step()
        return losses

    def train_step(self, plbl):
        return self.train_finish(plbl, *self.train_forward())

    def step(self, marks=None):
        """marks: optional list of 5 events bracketing the four parts"""
        host = self.host_marks = [0.0] * 5       # host clock at the same points: how long the ENQUEUE of each part takes
        self.weak, self.strong = self.batches[self._step_no % len(self.batches)]
        self._step_no += 1

        def rec(i):
            host[i] = time.perf_counter()
            if marks is not None:
                marks[i].record()
        rec(0)
        mp, am = self.plabel_begin_async()
        rec(1)
        if self.pipelined:
            out, teacher_lr = self.train_forward()
            rec(2)
            plbl = self.plabel_finish(mp, am)
            rec(3)
        else:
            plbl = self.plabel_finish(mp, am)
            rec(2)
            rec(3)
            out, teacher_lr = self.train_forward()
            # (marks 2..3 then bracket nothing; the forward is counted with the rest of the training step)
        losses = self.train_finish(plbl, out, teacher_lr)
        rec(4)
        return losses, plbl


def _cpu_info():
    """CPU model / sockets / physical cores of the host (lscpu), as BASELINE.md §3 asks"""
    import subprocess
    info = {"model": None, "sockets": None, "cores_per_socket": None, "threads_per_core": None, "logical": os.cpu_count()}
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        for line in out.splitlines():
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "Model name":
                info["model"] = v
            elif k == "Socket(s)":
                info["sockets"] = int(v)
            elif k == "Core(s) per socket":
                info["cores_per_socket"] = int(v)
            elif k == "Thread(s) per core":
                info["threads_per_core"] = int(v)
    except Exception:
        pass
    if info["sockets"] and info["cores_per_socket"]:
        info["physical_cores"] = info["sockets"] * info["cores_per_socket"]
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:
        usable = os.cpu_count() or 1
    # the box's CPU share: a cgroup quota does not show in the affinity mask (256 logical CPUs visible, 16 granted:
    # 256 torch threads on 16 CPUs ran the baseline 10x slower than 16 threads)
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: (t.split()[0], t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            txt = open(path).read().strip()
            if parse is not None:
                quota, period = parse(txt)
            else:
                quota, period = txt, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            if quota not in ("max", "-1") and int(period) > 0:
                usable = max(1, min(usable, int(int(quota) / int(period))))
                info["cgroup_cpu_quota"] = int(quota) / int(period)
                break
        except Exception:
            continue
    info["usable"] = usable
    return info


def cpu_baseline(cfg, size, threads, batch=2, reps=3):
    """The reference's CPU path, restated (oracle/), timed as BASELINE.md §3 prescribes: batch 2, one warm-up + `reps`
    timed repetitions (median) of (i) eval forward + the list / np.quantile IAS post-processing incl. the per-row
    threshold map (np.apply_along_axis) and (ii) one self-training step (teacher forward, student forward, 4-term loss,
    backward, Adam); (iii) the single-threaded host post-processing alone.  Per-image times, scaled to 1024x512 by
    the pixel ratio when a smaller sample size is asked for."""
    from oracle import deeplab_ref, ias_ref, losses_ref
    from hiast_amd.utils.registry.registries import MODEL
    h, w = size
    info = _cpu_info()
    cores = max(1, min(threads, info["usable"] or 1))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    sd = {k: v.detach().clone() for k, v in MODEL[cfg.model.type](cfg).state_dict().items()}
    x = torch.randn(batch, 3, h, w)
    paths = ["img_%d.png" % i for i in range(batch)]

    def generate():
        t0 = time.perf_counter()
        with torch.no_grad():
            logits, _, _ = deeplab_ref.segmentor_logits(x, sd)
            pp, lp = torch.softmax(logits, 1).max(1)
        t1 = time.perf_counter()
        st = ias_ref.IASState(C, 0.5, 0.9, 8.0)
        plbl = st.step(pp.numpy(), lp.numpy(), paths)
        for b in range(batch):   # the reference's per-pixel threshold map (np.apply_along_axis over rows, :74)
            np.apply_along_axis(lambda r: [st.class_threshold[e] for e in r], 1, lp.numpy()[b])
        t2 = time.perf_counter()
        return plbl, t1 - t0, t2 - t1

    def train(plbl):
        t0 = time.perf_counter()
        params = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and ".bn" not in k and "downsample.1" not in k)
                  for k, v in sd.items()}
        with torch.no_grad():
            zt = deeplab_ref.segmentor_logits(x, sd)[1]
        _, zs, _ = deeplab_ref.segmentor_logits(x, params, train=True)
        L = losses_ref.st_losses(zs, zt, torch.from_numpy(plbl.astype(np.int64)), (h, w), "ignored", dtype=torch.float32)
        sum(v for v in L.values() if torch.isfinite(v)).backward()
        opt = torch.optim.Adam([p for p in params.values() if p.requires_grad and p.grad is not None], lr=3e-6,
                               weight_decay=0.0005)
        opt.step()
        return time.perf_counter() - t0

    def say(msg):                      # progress on stderr: a harness that sees no output for minutes may take the run for hung
        print("[cpu_baseline] " + msg, file=sys.stderr, flush=True)

    say("%d torch threads on %s; warm-up ..." % (cores, info["model"]))
    plbl, _, _ = generate()            # warm-up (first-touch allocations, thread pools)
    train(plbl)
    gen, post, trn = [], [], []
    for r in range(reps):
        plbl, tf, tp = generate()
        gen.append(tf)
        post.append(tp)
        trn.append(train(plbl))
        say("repetition %d/%d: eval forward %.1f s, IAS post-processing %.1f s, training step %.1f s (batch %d, %dx%d)"
            % (r + 1, reps, tf, tp, trn[-1], batch, w, h))
    scale = (H * W) / float(h * w) / batch          # -> seconds per 1024x512 image
    med = lambda v: float(np.median(v))
    fwd, pst, tr = med(gen) * scale, med(post) * scale, med(trn) * scale
    total = fwd + pst + tr
    return {"value": 1.0 / total, "unit": "images/s", "cores": cores, "kind": "port",
            "cpu": info,
            "detail_s_per_image": {"eval_forward": fwd, "ias_post_processing_1_thread": pst, "train_step": tr},
            "repetitions": {"warmup": 1, "timed": reps, "statistic": "median", "batch": batch,
                            "eval_forward_s": gen, "ias_post_s": post, "train_step_s": trn},
            "sample": "batch of %d images %dx%d (per-image times x%.2f to 1024x512), 1 warm-up + %d timed repetitions, "
                      "median: eval forward %.2f s + IAS list/np.quantile/apply_along_axis post-processing %.2f s "
                      "(single-threaded, as in the reference) + HIAST training step %.2f s per image; torch %d threads "
                      "on %s (%s sockets x %s cores, %s usable logical CPUs)"
                      % (batch, w, h, (H * W) / float(h * w), reps, fwd, pst, tr, cores, info["model"], info["sockets"],
                         info["cores_per_socket"], info["usable"])}


class Watchdog:
    """A daemon THREAD of this rank (never a re-exec: the process has touched the GPU): `beat(stage)` after every finished
    stage / step; `limit` seconds without a beat -> every rank says on stderr where it stands (stage, step, collectives issued
    so far per communicator against the expected 6 / 208 / 3 per step), rank 0 writes ONE diagnostic JSON line to the stdout
    descriptor (value null, "error") and the process ends with exit code 3 (os._exit: a main thread parked inside a collective
    cannot be unwound).  torch.distributed.run then takes the other ranks down."""

    def __init__(self, limit, rank, world, json_out, args):
        import threading
        self.limit, self.rank, self.world, self.json_out, self.args = limit, rank, world, json_out, args
        self.stage, self.t_beat, self.t0, self.done = "start", time.monotonic(), time.monotonic(), False
        self.extra = {}
        if limit > 0:
            threading.Thread(target=self._run, name="hiast-bench-watchdog", daemon=True).start()

    def beat(self, stage, **extra):
        self.stage, self.t_beat = stage, time.monotonic()
        self.extra.update(extra)

    def stop(self):
        self.done = True

    def diagnostic(self):
        from hiast_amd.utils import comm
        return {"metric": "self-training images/sec (fwd+bwd+pseudo-label) at 1024x512", "value": None, "unit": "images/s",
                "n_gpus": self.world, "steps": self.args.steps, "warmup": self.args.warmup, "ms_per_step": None,
                "higher_is_better": True, "error": "watchdog: no progress for %.0f s" % (time.monotonic() - self.t_beat),
                "watchdog": {"rank": self.rank, "stage": self.stage, "seconds_since_start": time.monotonic() - self.t0,
                             "limit_s": self.limit, "backend": self.args.backend,
                             "collectives_issued": dict(comm.COUNTS), "collective_host_s": dict(comm.HOST_S),
                             "expected_per_step": {"gradient_buckets": 6, "syncbn_stat_all_reduces": 208,
                                                   "pseudo_label_aux_all_reduces": 3},
                             "dist_timeout_s": comm.timeout().total_seconds(), **self.extra}}

    def _run(self):
        while not self.done:
            time.sleep(1.0)
            if self.done or time.monotonic() - self.t_beat < self.limit:
                continue
            d = self.diagnostic()
            print("[bench watchdog] rank %d: %s" % (self.rank, json.dumps(d["watchdog"])), file=sys.stderr, flush=True)
            if self.rank == 0:
                try:
                    self.json_out.write(json.dumps(d) + "\n")
                    self.json_out.flush()
                except Exception:
                    pass
            os._exit(3)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (the reference's train.py does the
    same with mp.spawn, code/train.py:52-59,82) as ONE torch.distributed.run child — before this process has made any
    GPU call — pass its stdout (rank 0's JSON line) through, and exit with its return code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from hiast_amd.utils import comm
    env.setdefault("OMP_NUM_THREADS", str(max(1, comm.usable_cpus() // n)))     # the cgroup's CPU share, not the 256 visible
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    # stdout carries exactly ONE line (rank 0's JSON).  Native libraries write there too ([Gloo] connection notes, RCCL's
    # NCCL_DEBUG / "NCCL WARN" lines, MIOpen): file descriptor 1 is pointed at stderr for the life of the process and the
    # JSON line goes to a duplicate of the original descriptor.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    multi = world > 1
    if args.rehearse_dist:
        assert world == 1, "--rehearse-dist is the ONE-rank rehearsal of the N > 1 path"
        multi = True
        os.environ["HIAST_DIST_REHEARSAL"] = "1"        # (before anything of the package asks: utils/comm.py rehearsal())
        if "MASTER_PORT" not in os.environ:             # a free port of this host (the group has one rank: nobody else needs it)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
    if args.global_batch:
        assert args.global_batch % world == 0, "--global-batch %d does not divide over %d ranks" % (args.global_batch, world)
        args.batch = args.global_batch // world
    import __graft_entry__ as ge
    if rank == 0 and not os.path.exists(os.path.join(ROOT, "hiast_amd", "csrc", "libhiast_hip.so")):
        ge.build()
    assert torch.cuda.is_available(), "bench.py needs the MI355X; there is no CPU fallback"
    if args.same_device:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dog = Watchdog(args.watchdog_s, rank, world, json_out, args)
    from hiast_amd.utils import comm
    if multi:
        # (no device_id: communicators are then created lazily by ncclCommInitRank at the first collective of each group —
        # the oldest and most exercised path of torch's RCCL backend — instead of eagerly + ncclCommSplit for new groups)
        # timeout = HIAST_DIST_TIMEOUT_S (180 s): a collective one rank never joins raises instead of hanging for 10-30 minutes;
        # SyncBN sums and the histogram exchange get communicators of their own, beside DDP's (comm.setup inside)
        comm.init_process_group(args.backend)
        dog.beat("process group up")
    # HIAST_RESERVE_CUS=n (N > 1; HIAST_RESERVE_CUS_FORCE=1: also here): the main stream and every side stream leave n CUs alone
    cu_reserve = comm.apply_cu_reserve(world, device)
    torch.backends.cudnn.benchmark = bool(int(os.environ.get("HIAST_MIOPEN_FIND", "0")))

    cfg = make_cfg(2 if args.rehearse_dist else world, args.trainer, args.amp_dtype)      # (gpu_num > 1 => SyncBN layers)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):        # the package's progress prints ("%% freeze all BN layers" ...)
        from hiast_amd.utils import utils as _u
        _u.limit_cpu_threads()      # (N ranks per node: N OpenMP pools sized for the whole machine otherwise)
        hp = HotPath(cfg, device, rank, world, args.batch, multi)   # must not precede the ONE JSON line on stdout
    dog.beat("model built")
    timer = KernelTimer()
    timer.install()

    def sync():
        if multi:
            dist.barrier(device_ids=[local]) if args.backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    from hiast_amd import functional as HF
    for it in range(args.warmup):
        # the LAST warm-up step runs in the configuration of the first timed step (everything in order on the main stream,
        # whole-batch launches): the library's one-time algorithm search for those shapes stays out of the timed region,
        # and the caching allocator holds the whole-batch blocks that step asks for (with the serial step first and the
        # overlapped steps — half-batch blocks — after it, a timed run was seen to pay four fresh 537 MB device allocations,
        # 5 ms each, inside the first timed step)
        serial = it == args.warmup - 1 and args.warmup >= 2      # (a single warm-up step: the configuration of most timed steps)
        hp.use_side = not serial
        HF.enable_wgrad_overlap(not serial)
        hp.step()
        if multi or it == 0:
            torch.cuda.synchronize()        # (the first step of a rank compiles nothing but loads ~200 code objects; at N > 1
        dog.beat("warm-up step %d of %d" % (it + 1, args.warmup))      # every warm-up step is a checkpoint of the watchdog)
    hp.use_side = True
    HF.enable_wgrad_overlap(True)
    sync()
    # fp16: the dynamic loss scale starts at 2^16 (apex) and is halved on every overflow; a step that overflows skips its
    # optimiser update.  No such step may sit in the timed region: extra (untimed) steps until six in a row have been APPLIED
    # at the current scale, and the applied-step counter is compared again after the timed steps (reported in the JSON).
    settle, clean = 0, 0
    if hp.scaler is not None:
        while settle < 40 and clean < 6:        # until six consecutive steps have been applied at the current scale
            before = hp.opt.applied_steps()
            hp.step()
            dog.beat("loss-scale settle step %d" % (settle + 1))
            settle += 1
            clean = clean + 1 if hp.opt.applied_steps() == before + 1 else 0
        sync()
    applied_before = hp.opt.applied_steps() if hp.scaler is not None else None

    def marker():
        """a uniquely named tiny kernel (hiast::confusion_kernel) that brackets the timed region in a
        rocprofv3 kernel trace; tools/trace_summary.py keeps only what lies between two markers"""
        from hiast_amd import kernels as K
        z = torch.zeros(64, dtype=torch.int64, device=device)
        K.confusion_hist(z, z, 2)
        torch.cuda.synchronize()
    marker()
    timer.on = True
    from hiast_amd.utils import comm as _comm
    coll_before, coll_host_before = dict(_comm.COUNTS), dict(_comm.HOST_S)
    # phases are timed with HIP events on the launch stream: no host synchronisation inside the timed region (the only
    # blocking point is the histogram read-back the IAS threshold update needs), so consecutive steps pipeline
    marks, host_parts = [], []
    from hiast_amd import functional as HF
    t0 = time.perf_counter()
    for it in range(args.steps):
        timer.on = it == 0              # per-launch events cost ~5 us each (460 per step): the first timed step only
        hp.use_side = not timer.on      # ... and on those steps nothing runs beside the timed kernels (teacher forward,
        HF.enable_wgrad_overlap(not timer.on)   # pseudo-label forward and weight gradients on the main stream, one launch
                                                # sequence per forward), so the per-launch durations are not stretched by
                                                # co-running work
        e = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        last_losses, _ = hp.step(e)
        marks.append(e)
        if not timer.on:
            host_parts.append([1e3 * (hp.host_marks[i + 1] - hp.host_marks[i]) for i in range(4)])
        dog.beat("timed step %d of %d enqueued" % (it + 1, args.steps))
    sync()
    dog.beat("timed region done")
    elapsed = time.perf_counter() - t0
    # collectives issued per step and communicator inside the timed region (host-side counters; the reducer's own bucket count)
    collectives = None
    if multi:
        collectives = {"syncbn_stat_all_reduces": (_comm.COUNTS["stat"] - coll_before["stat"]) / float(args.steps),
                       "pseudo_label_aux_all_reduces": (_comm.COUNTS["aux"] - coll_before["aux"]) / float(args.steps),
                       "gradient_buckets": None,
                       # host time inside the all_reduce / wait calls of each communicator (DESIGN §7 budgets 208 x 20-40 us on
                       # the statistics group); DDP's bucket reduces are issued by the reducer's C++ hooks: not timed here
                       "host_ms_per_step": {k: 1e3 * (_comm.HOST_S[k] - coll_host_before[k]) / float(args.steps)
                                            for k in ("stat", "aux")},
                       "cu_reserve": cu_reserve,
                       "expected": {"syncbn_stat_all_reduces": 2 * sum(1 for m in hp.model.modules()
                                                                       if isinstance(m, torch.nn.SyncBatchNorm)),
                                    "pseudo_label_aux_all_reduces": 3}}
        try:
            collectives["gradient_buckets"] = int(hp.model._get_ddp_logging_data().get("num_buckets_reduced"))
        except Exception:
            pass
    # phase split from the FIRST timed step only: it runs every part on the main stream (see above); in the other steps
    # the pseudo-label forward runs beside the training forwards and the marks of the main stream do not separate them
    ser = marks[:1]
    if hp.pipelined:    # pseudo-label pass = [0,1] + [2,3]; training step = [1,2] + [3,4]
        t_pl = sum(e[0].elapsed_time(e[1]) + e[2].elapsed_time(e[3]) for e in ser) * 1e-3 * args.steps
        t_tr = sum(e[1].elapsed_time(e[2]) + e[3].elapsed_time(e[4]) for e in ser) * 1e-3 * args.steps
    else:
        t_pl = sum(e[0].elapsed_time(e[2]) for e in ser) * 1e-3 * args.steps
        t_tr = sum(e[3].elapsed_time(e[4]) for e in ser) * 1e-3 * args.steps
    timer.on = False
    marker()
    # unprofiled device-side gap at the step boundary: time between the event behind the last launch of step i (after the
    # EMA update / buffer copy) and the event in front of the first launch of step i + 1, on the main stream.  If the host
    # enqueues ahead of the device the two events complete back to back (~0); a host that arrives late shows as a gap.
    gaps = [marks[i][4].elapsed_time(marks[i + 1][0]) for i in range(1, len(marks) - 1)]      # (step 0 is the serial one)
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # after the timed region: the last step's losses (mean over ranks) and, at N > 1, whether the replicas still agree —
    # IAS thresholds as float64 bit patterns and a checksum over every parameter must be identical on all ranks (a rank
    # that missed an all-reduce, or reduced on the wrong communicator, drifts within one step)
    names = list(last_losses.keys())
    lv = torch.stack([torch.mean(last_losses[k]).double() for k in names])
    ranks_agree = None
    if multi:
        dist.all_reduce(lv)
        lv /= world
        chk = torch.cat([torch.from_numpy(hp.thr.view(np.int64).copy()).to(device),
                         torch.stack([p.detach().double().sum() for p in hp.model.parameters()]).view(torch.int64)])
        both = torch.stack([chk, -chk])
        dist.all_reduce(both, op=dist.ReduceOp.MAX)         # max(x) == -max(-x) on every element <=> all ranks equal
        ranks_agree = bool(torch.equal(both[0], -both[1]))
    final_losses = dict(zip(names, [float(v) for v in lv.cpu()]))
    skipped = None
    if hp.scaler is not None:       # optimiser updates skipped inside the timed region (must be 0: see the warm-up above)
        skipped = args.steps - (hp.opt.applied_steps() - applied_before)

    if rank == 0:
        imgs = world * args.batch * args.steps
        out = {
            "metric": "self-training images/sec (fwd+bwd+pseudo-label) at 1024x512",
            "value": imgs / elapsed, "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None, "dtype": args.amp_dtype,
            "dtype_detail": "16-bit MFMA with fp32 accumulate everywhere: %s in the training step, split-bf16 planes "
                            "(hi*hi+lo*hi+hi*lo, fp32-class) in the pseudo-label forward incl. its ASPP head; losses / "
                            "softmax / thresholds fp32 + integer"
                            % ("IEEE fp16 operands with apex-style dynamic loss scaling — the reference's own training "
                               "arithmetic (apex O1)" if args.amp_dtype == "fp16" else
                               "plain bf16 operands (no loss scaling; the reference trains under apex O1 = fp16: "
                               "--amp-dtype fp16 runs that on the same kernels)"),
            "loss_scale": ({"final": float(hp.scaler.get_scale()), "settle_steps_before_timing": settle,
                            "optimizer_steps_skipped_in_timed_region": skipped} if hp.scaler is not None else None),
            "data": "synthetic",
            "data_detail": "two seeded synthetic batches per rank, alternating step by step (white noise; smoothed noise at "
                           "another contrast), resident in HBM",
            "config": {"workload": "configs[2] self-training round (%s, region-adaptive reg on) bs=%d/GPU @1024x512 "
                                   "+ configs[1] IAS pseudo-label pass on the same batch" % (args.trainer, args.batch),
                       "images_per_gpu_per_step": args.batch, "num_classes": C,
                       "batch_semantics": ("reference_bs%d: global batch split over the ranks + SyncBN (code/train.py:52-53)"
                                           % args.global_batch if args.global_batch else "per-GPU batch (weak scaling)"),
                       "parallelism": "dp1-rehearsal" if args.rehearse_dist else ("dp%d" % world if world > 1 else "single"), "cu_reserve": cu_reserve},
            "phases_ms": {"pseudo_label": 1e3 * t_pl / args.steps, "train_step": 1e3 * t_tr / args.steps,
                          "note": "of the first timed step, which runs every part in order on one stream (per-launch "
                                  "events); the other steps overlap the parts on four streams"},
            # host time spent ENQUEUING each part (steps without per-launch events; the third entry includes the wait
            # for the histogram): the sum must stay below ms_per_step or the step is launch-bound
            "final_losses": final_losses, "ranks_agree": ranks_agree, "collectives_per_step": collectives,
            "step_boundary_gap_ms": ({"mean": float(np.mean(gaps)), "max": float(np.max(gaps)), "steps": len(gaps),
                                      "note": "HIP events on the main stream behind the last launch of a step and in front "
                                              "of the first launch of the next (unprofiled run)"} if gaps else None),
            "host_enqueue_ms": dict(zip(["plabel_fwd_pass1", "train_forwards", "hist_wait_thresholds_pass2",
                                         "loss_bwd_adam_ema"] if hp.pipelined else
                                        ["plabel_fwd_pass1", "hist_wait_thresholds_pass2", "-", "train_step"],
                                        [float(v) for v in np.mean(np.array(host_parts), axis=0)])) if host_parts else None,
        }
        groups = timer.summary()
        groups_all = list(groups)
        if groups:
            sampled = 1                                  # steps on which launches were timed
            aspp = [g for g in groups if g[0][0] == "aspp2_fwd"]
            wg = [g for g in groups if g[0][0] == "wgrad_group"]      # (reported beside the convolution groups, below)
            groups = [g for g in groups if g[0][0] == "igemm"]       # (the ASPP GEMM is also counted in the igemm groups)
            key, avg_ms, n, _tot = groups[0]            # dominant hand-written launch group of the step
            out["roofline"] = roofline_of(key, avg_ms, n, sampled)
            others = [roofline_of(k, a, c, sampled) for k, a, c, _ in groups[1:4] + wg[:1] + aspp[:2]]
            out["roofline_other"] = [{kk: o[kk] for kk in ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_issued",
                                                            "frac_algorithmic_vs_dense", "traffic", "traffic_source",
                                                            "avg_launch_ms", "launches_per_step", "note") if kk in o}
                                     for o in others]
        out.update(step_fractions(groups_all, 1e3 * elapsed / args.steps, 1e3 * (t_pl + t_tr) / args.steps))
        if not args.no_cpu_baseline and world == 1:
            dog.stop()          # (the CPU baseline reports its own progress on stderr)
            try:
                with contextlib.redirect_stdout(sys.stderr):
                    out["cpu_baseline"] = cpu_baseline(cfg, tuple(args.cpu_size), args.cpu_threads, reps=args.cpu_reps)
            except Exception as e:      # the baseline must never take the measurement down
                out["cpu_baseline"] = {"error": repr(e)}
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    dog.stop()
    if multi:
        dist.barrier(device_ids=[local]) if args.backend == "nccl" else dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
