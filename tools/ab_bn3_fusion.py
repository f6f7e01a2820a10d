"""A/B of K9e' (hiast_xconv_dgrad_gated_bn_stats): python tools/ab_bn3_fusion.py — the fused launch against the two it replaces
(gated data gradient on xconv + bn3's backward statistics pass) on the layer3 shape of the bench (B = 8, 64 x 128, 256 -> 1024)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from hiast_amd import kernels as K  # noqa: E402
from ab_igemm import timeit  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, H, W, Kc, N = 8, 64, 128, 256, 1024
M = B * H * W
for dt in (torch.float16, torch.bfloat16):
    dy = torch.randn(B, H, W, Kc, device=dev).to(dt)
    w = torch.randn(Kc, N, 1, 1, device=dev) * (2.0 / Kc) ** 0.5
    wpt = K.pack_conv_weight(w, K.fmt_of(dy), transpose=True)
    res = torch.randn(B, H, W, N, device=dev).to(dt)
    gate = torch.randint(0, 256, (M, N // 8), dtype=torch.uint8, device=dev)
    bx = torch.randn(B, H, W, N, device=dev).to(dt)
    bmask = torch.randint(0, 256, (M, N // 8), dtype=torch.uint8, device=dev)
    sm = torch.zeros(N, device=dev)
    si = torch.ones(N, device=dev)
    t_f = timeit(lambda: K.xconv_dgrad_gated_bn_stats(dy, wpt, res, gate, bx, bmask, sm, si), n=40)
    t_d = timeit(lambda: K.igemm_bn_act(dy, wpt, 1, None, res, False, 1, 1, res_gate=gate), n=40)
    dx = K.igemm_bn_act(dy, wpt, 1, None, res, False, 1, 1, res_gate=gate)
    t_s = timeit(lambda: K.bn_nhwc_bwd_stats(dx.permute(0, 3, 1, 2), bmask, bx.permute(0, 3, 1, 2), None, None, sm, si, 3), n=40)
    mb_f = (M * (Kc + 3 * N) * 2 + 2 * M * N // 8) / 1e6
    print("%s: fused %.1f us (%.0f MB -> %.2f TB/s) | gated data gradient %.1f us + statistics pass %.1f us = %.1f us"
          % (str(dt)[6:], t_f * 1e3, mb_f, mb_f / t_f / 1e3, t_d * 1e3, t_s * 1e3, (t_d + t_s) * 1e3), flush=True)
