"""Timing of the channels-last BatchNorm passes on the layer3 shapes, beside torch's copy / add (HBM reference rates).
Run once per build (HIAST_LIB=...) to A/B two libraries on one box."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hiast_amd import kernels as K      # noqa: E402


def timeit(fn, n=40, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2] * 1e3


def main():
    dev = torch.device("cuda:0")
    B, H, W = 8, 64, 128
    M = B * H * W
    print("library:", os.environ.get("HIAST_LIB", "in-tree"))
    for C in ([int(a) for a in sys.argv[1:]] or (256, 1024, 512, 2048)):
        cl = torch.channels_last
        x = torch.randn(B, C, H, W, device=dev).bfloat16().contiguous(memory_format=cl)
        r = torch.randn(B, C, H, W, device=dev).bfloat16().contiguous(memory_format=cl)
        dy = torch.randn(B, C, H, W, device=dev).bfloat16().contiguous(memory_format=cl)
        g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
        rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev)
        mb = M * C * 2 / 1e6
        nblk = M // 256
        partial = torch.randn(nblk, C, 2, device=dev).abs() * 100
        partial[:, :, 1] += 1e4
        def rate(t, nbytes):
            return "%6.1f us %5.2f TB/s" % (t, nbytes / t)
        row = "C=%4d (%5.1f MB/tensor) |" % (C, mb)
        y = torch.empty_like(x)
        row += " torch copy " + rate(timeit(lambda: y.copy_(x)), 2 * mb) + " |"
        row += " torch add " + rate(timeit(lambda: torch.add(x, r, out=y)), 3 * mb) + " |"
        print(row, flush=True)
        row = "    fwd apply+relu " + rate(timeit(lambda: K.bn_nhwc_apply_partial(x, None, g, b, rm, rv, partial, float(M), 0.1, 1e-5, True)), 2 * mb)
        row += " | fwd apply+res+relu+mask " + rate(timeit(lambda: K.bn_nhwc_apply_partial(x, r, g, b, rm, rv, partial, float(M), 0.1, 1e-5, True, True)), 3 * mb + mb / 16)
        print(row, flush=True)
        out = K.bn_nhwc_apply_partial(x, r, g, b, rm, rv, partial, float(M), 0.1, 1e-5, True, True)
        sm, si, mask = out[1], out[2], out[3]
        row = "    bwd stats gate2 " + rate(timeit(lambda: K.bn_nhwc_bwd_stats(dy, None, x, g, b, sm, si, 2)), 2 * mb)
        row += " | gate3 " + rate(timeit(lambda: K.bn_nhwc_bwd_stats(dy, mask, x, g, b, sm, si, 3)), 2 * mb + mb / 16)
        sums = K.bn_nhwc_bwd_stats(dy, None, x, g, b, sm, si, 2)
        row += " | bwd apply gate2 " + rate(timeit(lambda: K.bn_nhwc_bwd_apply(dy, None, x, g, b, sm, si, sums, float(M), 2, False, False)), 3 * mb)
        row += " | gate3 " + rate(timeit(lambda: K.bn_nhwc_bwd_apply(dy, mask, x, g, b, sm, si, sums, float(M), 3, False, False)), 3 * mb + mb / 16)
        print(row, flush=True)


if __name__ == "__main__":
    main()
