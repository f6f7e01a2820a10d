"""Per-kernel timing of the HIP library on the MI355X (HIP events on torch's current stream, inputs of the
bench workload: B images of 1024x512 -> 64x128 head maps).  Usage: python tools/bench_kernels.py [B]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hiast_amd import kernels as K  # noqa: E402


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2], ms[0]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["aspp", "plabel", "loss", "upsample", "ema"]
    dev = torch.device("cuda")
    C, Cin, h, w, H, W = 19, 2048, 64, 128, 512, 1024
    dil = (6, 12, 18, 24)
    torch.manual_seed(0)
    if "aspp" in which:
        x = torch.randn(B, Cin, h, w, device=dev)
        ws = [torch.randn(C, Cin, 3, 3, device=dev) * 0.01 for _ in range(4)]
        bs = [torch.randn(C, device=dev) * 0.1 for _ in range(4)]
        dy = torch.randn(B, C, h, w, device=dev)
        wpack = K.aspp_pack_weights(ws, bs)
        wsp = K.aspp_workspace(B, Cin, h, w, C, dev)
        gf = 2.0 * h * w * C * Cin * 36 * B / 1e9
        for name, fn in (("aspp_pack", lambda: K.aspp_pack_weights(ws, bs)),
                         ("aspp_fwd", lambda: K.aspp_fwd(x, wpack, C, dil, wsp)),
                         ("aspp_bwd_data", lambda: K.aspp_bwd_data(dy, wpack, Cin, dil)),
                         ("aspp_bwd_weight", lambda: K.aspp_bwd_weight(x, dy, dil, wsp))):
            med, best = timeit(fn)
            print("%-18s B=%d  median %8.3f ms  best %8.3f ms  %7.2f TFLOP/s (algorithmic, fp32)" %
                  (name, B, med, best, gf / med if "pack" not in name else 0))
        del x, dy
    if "aspp2" in which:
        ws = [torch.randn(C, Cin, 3, 3, device=dev) * 0.01 for _ in range(4)]
        bs = [torch.randn(C, device=dev) * 0.1 for _ in range(4)]
        dy = torch.randn(B, C, h, w, device=dev)
        wt, wd, bias = K.aspp2_pack_weights(ws, bs)
        gf = 2.0 * h * w * C * Cin * 36 * B / 1e9
        x32 = torch.randn(B, Cin, h, w, device=dev).contiguous(memory_format=torch.channels_last)
        x16 = x32.bfloat16()
        wsf = K.aspp2_workspace(B, Cin, h, w, C, False, dev)
        wsb = K.aspp2_workspace(B, Cin, h, w, C, True, dev)
        for name, fn in (("aspp2_pack", lambda: K.aspp2_pack_weights(ws, bs)),
                         ("aspp2_fwd fp32split", lambda: K.aspp2_fwd(x32, wt, bias, dil, wsf)),
                         ("aspp2_fwd bf16", lambda: K.aspp2_fwd(x16, wt, bias, dil, wsf)),
                         ("aspp2_bwd dx only", lambda: K.aspp2_bwd(x16, dy, wd, dil, True, False, wsb)),
                         ("aspp2_bwd dw only", lambda: K.aspp2_bwd(x16, dy, wd, dil, False, True, wsb)),
                         ("aspp2_bwd both", lambda: K.aspp2_bwd(x16, dy, wd, dil, True, True, wsb))):
            med, best = timeit(fn)
            print("%-20s B=%d  median %8.3f ms  best %8.3f ms  %7.2f TFLOP/s (algorithmic)" %
                  (name, B, med, best, gf / med * (2 if "both" in name else 1) if "pack" not in name else 0), flush=True)
        del x32, x16, dy
    if "igemm" in which:
        Bn, Hh, Ww = B, 64, 128
        cases = [("l3.conv2 3x3 d2", 256, 256, 9, 2, False), ("l3.conv1 1x1", 1024, 256, 1, 1, False),
                 ("l3.conv3 1x1+res", 256, 1024, 1, 1, True), ("l4.conv2 3x3 d4", 512, 512, 9, 4, False),
                 ("l4.conv1 1x1", 2048, 512, 1, 1, False), ("l4.conv3 1x1+res", 512, 2048, 1, 1, True),
                 ("l2.conv2 3x3", 128, 128, 9, 1, False), ("l2.conv3 1x1+res", 128, 512, 1, 1, True)]
        for name, Cin_, Cout_, taps, dl, has_res in cases:
            kk = 3 if taps == 9 else 1
            wt_ = torch.randn(Cout_, Cin_, kk, kk, device=dev) * (2.0 / (Cin_ * taps)) ** 0.5
            bn = torch.nn.BatchNorm2d(Cout_).to(dev).eval()
            x32 = torch.randn(Bn, Hh, Ww, Cin_, device=dev)
            gf = 2.0 * Bn * Hh * Ww * Cin_ * Cout_ * taps / 1e9
            row = "%-18s %6.1f GFLOP |" % (name, gf)
            for PL in (2, 1):
                xp = (K.split_planes(x32.view(-1, Cin_)).view(Bn, Hh, Ww, 2 * Cin_) if PL == 2 else x32.bfloat16())
                wp = K.pack_conv_weight(wt_, PL)
                res = torch.randn(Bn, Hh, Ww, PL * Cout_, device=dev).bfloat16() if has_res else None
                med, best = timeit(lambda: K.igemm_bn_act(xp, wp, PL, bn, res, True, 1, dl), n=20, warm=5)
                row += " PL%d %7.3f ms %6.0f TF/s(alg) |" % (PL, med, gf / med)
            # library: torch conv2d bf16 (MIOpen), NCHW
            xn = x32.permute(0, 3, 1, 2).contiguous().bfloat16()
            wb = wt_.bfloat16()
            med, best = timeit(lambda: torch.nn.functional.conv2d(xn, wb, None, 1, dl if taps == 9 else 0, dl), n=20, warm=5)
            row += " miopen bf16 conv only %7.3f ms" % med
            print(row, flush=True)
    if "xconv" in which:
        # the HBM-bound 1x1 launches of layer3 (K9e xconv.hip vs the tile kernel, HIAST_XCONV=0): algorithmic bytes / time
        Bn, Hh, Ww = B, 64, 128
        M = Bn * Hh * Ww
        x = torch.randn(Bn, Hh, Ww, 256, device=dev).bfloat16()
        wp = K.pack_conv_weight(torch.randn(1024, 256, 1, 1, device=dev) * 0.05, 1)
        res = torch.randn(Bn, Hh, Ww, 1024, device=dev).bfloat16()
        bits = torch.randint(0, 256, (M, 128), device=dev, dtype=torch.uint8)
        bn = torch.nn.BatchNorm2d(1024).to(dev).eval()
        mb = lambda *t: sum(v.numel() * v.element_size() for v in t) / 1e6
        out_mb = M * 1024 * 2 / 1e6
        variants = [("conv3 student (stats)", lambda: K.igemm_bn_act(x, wp, 1, None, None, False, want_stats=True), mb(x) + out_mb),
                    ("conv3+res+relu teacher", lambda: K.igemm_bn_act(x, wp, 1, bn, res, True), mb(x, res) + out_mb),
                    ("conv1 dgrad gated res", lambda: K.igemm_bn_act(x, wp, 1, None, res, False, res_gate=bits), mb(x, res, bits) + out_mb),
                    ("plain", lambda: K.igemm_bn_act(x, wp, 1, None, None, False), mb(x) + out_mb)]
        for name, fn, mbytes in variants:
            row = "xconv %-24s %6.0f MB |" % (name, mbytes)
            for flag in ("1", "0"):
                os.environ["HIAST_XCONV"] = flag
                med, best = timeit(fn, n=30, warm=5)
                row += " %s %7.3f ms (best %6.3f) %5.2f TB/s |" % ("xconv" if flag == "1" else "tile ", med, best, mbytes / med / 1e3)
            os.environ.pop("HIAST_XCONV", None)
            print(row, flush=True)
    if "wgrad" in which:
        Bn, Hh, Ww = B, 64, 128
        for name, Cin_, Cout_, k, dl in (("l3.conv1", 1024, 256, 1, 1), ("l3.conv2", 256, 256, 3, 2), ("l3.conv3", 256, 1024, 1, 1),
                                        ("l4.conv1", 2048, 512, 1, 1), ("l4.conv2", 512, 512, 3, 4), ("l4.conv3", 512, 2048, 1, 1)):
            x = torch.randn(Bn, Hh, Ww, Cin_, device=dev).bfloat16()
            dy = torch.randn(Bn, Hh, Ww, Cout_, device=dev).bfloat16()
            gf = 2.0 * Bn * Hh * Ww * Cin_ * Cout_ * k * k / 1e9
            med, best = timeit(lambda: K.conv_wgrad_nhwc(dy, x, k, 1, dl), n=10, warm=3)
            xl, dyl = x.permute(0, 3, 1, 2), dy.permute(0, 3, 1, 2)
            wl = torch.empty(Cout_, Cin_, k, k, device=dev, dtype=torch.bfloat16)
            pad = dl if k == 3 else 0
            med2, _ = timeit(lambda: torch.ops.aten.convolution_backward(dyl, xl, wl, None, (1, 1), (pad, pad), (dl, dl), False,
                                                                         (0, 0), 1, (False, True, False)), n=10, warm=3)
            print("wgrad %-9s %6.1f GFLOP | own %7.3f ms %5.0f TF/s | miopen %7.3f ms %5.0f TF/s" %
                  (name, gf, med, gf / med, med2, gf / med2), flush=True)
    if "plabel" in which:
        z = torch.randn(B, C, h, w, device=dev) * 3
        med, best = timeit(lambda: K.plabel_pass1(z, H, W))
        mp, am, hist = K.plabel_pass1(z, H, W)
        byt = B * (C * h * w * 4 + H * W * 5) / 1e9
        print("%-18s B=%d  median %8.3f ms  best %8.3f ms  %7.1f GB/s (logits in + prob/label out)" % ("plabel_pass1", B, med, best, byt / med * 1e3))
        thr = torch.full((C,), 0.9, device=dev)
        med, best = timeit(lambda: K.plabel_pass2(mp, am, thr, C))
        byt = B * H * W * 6 / 1e9
        print("%-18s B=%d  median %8.3f ms  best %8.3f ms  %7.1f GB/s" % ("plabel_pass2", B, med, best, byt / med * 1e3))
    if "loss" in which:
        z = torch.randn(B, C, h, w, device=dev) * 3
        zt = torch.randn(B, C, h, w, device=dev) * 3
        pl = torch.randint(0, C, (B, H, W), device=dev, dtype=torch.uint8)
        pl[torch.rand(B, H, W, device=dev) < 0.4] = 255
        ws_ = K.st_loss_workspace(B, C, h, w, H, W, dev)
        coef = torch.tensor([1, .1, 1, .5], device=dev)
        med, best = timeit(lambda: K.st_loss_fwd(z, zt, pl, H, W, "ignored", ws_))
        byt = B * (2 * C * h * w * 4 + H * W) / 1e9
        print("%-18s B=%d  median %8.3f ms  best %8.3f ms  %7.1f GB/s (algorithmic)" % ("st_loss_fwd", B, med, best, byt / med * 1e3))
        sums = K.st_loss_fwd(z, zt, pl, H, W, "ignored", ws_)
        med, best = timeit(lambda: K.st_loss_bwd(z, zt, pl, H, W, "ignored", sums, coef, ws_))
        byt = B * (3 * C * h * w * 4 + H * W) / 1e9
        print("%-18s B=%d  median %8.3f ms  best %8.3f ms  %7.1f GB/s (algorithmic)" % ("st_loss_bwd", B, med, best, byt / med * 1e3))
    if "upsample" in which:
        z = torch.randn(B, C, h, w, device=dev)
        med, best = timeit(lambda: K.upsample_bilinear_ac_fwd(z, H, W))
        byt = B * C * (h * w + H * W) * 4 / 1e9
        print("%-18s B=%d  median %8.3f ms  best %8.3f ms  %7.1f GB/s" % ("upsample_fwd", B, med, best, byt / med * 1e3))
        g = torch.randn(B, C, H, W, device=dev)
        med, best = timeit(lambda: K.upsample_bilinear_ac_bwd(g, h, w))
        print("%-18s B=%d  median %8.3f ms  best %8.3f ms  %7.1f GB/s" % ("upsample_bwd", B, med, best, byt / med * 1e3))
    if "ema" in which:
        n = 44_000_000
        e = [torch.randn(n, device=dev)]
        p = [torch.randn(n, device=dev)]
        plan = K.EmaPlan(e, p)
        med, best = timeit(lambda: K.ema_update(plan, 0.999))
        print("%-18s n=%d median %8.3f ms  best %8.3f ms  %7.1f GB/s" % ("ema_update", n, med, best, n * 12 / 1e9 / med * 1e3))


if __name__ == "__main__":
    main()
