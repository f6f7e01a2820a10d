"""A/B of the training stem (K9k): python tools/ab_stem_train.py — hiast_stem_train_fwd / hiast_stem_wgrad against the
library path they replace (channels-last copy + cast + MIOpen convolution + statistics pass; MIOpen weight gradient)."""
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
B, H, W = 8, 512, 1024
x = torch.randn(B, 3, H, W, device=dev)
w = torch.randn(64, 3, 7, 7, device=dev) * 0.05
conv = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False).to(dev)
for dt in (torch.float16, torch.bfloat16):
    fmt = K.FMT_FP16 if dt == torch.float16 else K.FMT_BF16

    def lib_fwd():
        with torch.autocast("cuda", dtype=dt):
            y = conv(x.contiguous(memory_format=torch.channels_last)).contiguous(memory_format=torch.channels_last)
        return y, K.bn_nhwc_stats(y)
    t_lib = timeit(lib_fwd, n=20)
    t_own = timeit(lambda: K.stem_train_fwd(x, w, fmt), n=20)
    y, _ = K.stem_train_fwd(x, w, fmt)
    dy = torch.randn_like(y)
    xl = x.contiguous(memory_format=torch.channels_last).to(dt)
    dyl = dy.permute(0, 3, 1, 2)
    wl = w.to(dt)
    t_libw = timeit(lambda: torch.ops.aten.convolution_backward(dyl, xl, wl, None, (2, 2), (3, 3), (1, 1), False, (0, 0), 1,
                                                                 (False, True, False)), n=20)
    t_ownw = timeit(lambda: K.stem_wgrad(x, dy), n=20)
    print("%s  forward: library copy+cast+conv+stats %.1f us | own %.1f us   weight gradient: library %.1f us | own %.1f us"
          % (str(dt)[6:], t_lib * 1e3, t_own * 1e3, t_libw * 1e3, t_ownw * 1e3), flush=True)
