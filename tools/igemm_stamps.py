"""In-kernel cycle breakdown of the igemm main loop (diagnostic build only):
    cd hiast_amd/csrc && hipcc ... -DIG_STAMP -c igemm.hip ... -o libhiast_hip_stamp.so
    HIAST_LIB=.../libhiast_hip_stamp.so python3 tools/igemm_stamps.py
Per wave: cycles (s_memtime) summed over the k-steps in  wait+barrier | DMA issue | first fragment reads | MFMAs + reads."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hiast_amd import kernels as K      # noqa: E402
from hiast_amd import _lib              # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B, H, W = 8, 64, 128
    raw = ctypes.CDLL(_lib.LIB_PATH)
    cases = [("l3.conv2 3x3 d2", 256, 256, 9, 2, False), ("l3.conv1 1x1", 1024, 256, 1, 1, False),
             ("l3.conv3 1x1+res", 256, 1024, 1, 1, True), ("l4.conv2 3x3 d4", 512, 512, 9, 4, False)]
    for name, ci, co, taps, dl, has_res in cases:
        kk = 3 if taps == 9 else 1
        wt = torch.randn(co, ci, kk, kk, device=dev) * (2.0 / (ci * taps)) ** 0.5
        x32 = torch.randn(B, H, W, ci, device=dev)
        for PL in (1, 2):
            xp = K.split_planes(x32.view(-1, ci)).view(B, H, W, 2 * ci) if PL == 2 else x32.bfloat16()
            wp = K.pack_conv_weight(wt, PL)
            bn = torch.nn.BatchNorm2d(co).to(dev).eval()
            res = torch.randn(B, H, W, PL * co, device=dev).bfloat16() if has_res else None
            os.environ["HIAST_XCONV"] = "0"
            for _ in range(5):
                K.igemm_bn_act(xp, wp, PL, bn, res, True, 1, dl)
            torch.cuda.synchronize()
            dbuf = torch.zeros(1024 * 8 * 8 + 32 * 64, dtype=torch.int32, device=dev)
            rc = raw.hiast_igemm_debug_stamps(ctypes.c_void_p(dbuf.data_ptr()))
            assert rc == 0
            buf = dbuf.cpu().numpy().view(np.uint32)
            st = buf[:1024 * 64].reshape(1024, 8, 8).astype(np.float64)
            nblk = (B * H * W // 256) * (co // 256)
            st = st[:min(nblk, 1024)]
            nk = st[0, 0, 5]
            per = st[:, :, :4] / nk                      # cycles per k-step
            tot = st[:, :, 4] / nk
            print("%-16s PL%d nk %3d | per k-step (100 MHz ticks x?): wait+barrier %7.1f | dma issue %6.1f | first reads %6.1f | mfma+reads %7.1f"
                  " | loop total %7.1f  (min/max over waves of total: %.1f / %.1f)" %
                  (name, PL, nk, per[..., 0].mean(), per[..., 1].mean(), per[..., 2].mean(), per[..., 3].mean(), tot.mean(),
                   tot.min(), tot.max()), flush=True)
            tl = buf.reshape(-1, 64)[1024:1032, :12].astype(np.int64)
            base = tl[:, 0].min()
            for w in range(8):
                print("    k-step 10, block 5, wave %d: top %5d | after barrier %5d | first reads done %5d | tiles %s | end %5d" %
                      (w, tl[w, 0] - base, tl[w, 1] - base, tl[w, 2] - base, " ".join("%5d" % (v - base) for v in tl[w, 3:11]),
                       tl[w, 11] - base), flush=True)
            ep = buf.reshape(-1, 64)[1032:1048, :10].astype(np.int64)
            for w in (0, 4, 7, 8, 15):
                if ep[w, 0] == 0:
                    continue
                b0_ = ep[w, 0]
                print("    block %s wave %d: kernel start 0 | loop start %6d | epilogue start %6d | chunks done %s" %
                      ("5" if w < 8 else "600", w & 7, ep[w, 1] - b0_, ep[w, 2] - b0_, " ".join("%6d" % (v - b0_) for v in ep[w, 3:8] if v)), flush=True)
            # spread of block start times (ticks) to see rounds
            b0 = st[:, 0, 6]
            print("    block start spread: %.0f ticks; wave 0 vs wave 7 wait: %.1f / %.1f" %
                  (b0.max() - b0.min(), per[:, 0, 0].mean(), per[:, 7, 0].mean()), flush=True)


if __name__ == "__main__":
    main()
