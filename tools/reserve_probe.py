"""Does a CU reserve give a collective's kernel a compute unit?  

At N > 1 the main stream waits on each of 208 chained [C,2] SyncBN all-reduces while the side streams (pseudo-label sub-batches,
EMA teacher) keep every CU busy with 60-160 us tile-kernel blocks (160 KiB of LDS, 2 x 256 registers: nothing co-resides).  An
RCCL kernel then queues until a block retires.  This probe measures exactly that on ONE GPU: a saturating sequence of layer3
3x3 tile-kernel launches on a "work" stream, and beside it a tiny probe kernel (one 256-thread block, as a small RCCL
all-reduce kernel needs one CU) launched on its own stream every ~200 us; the probe's latency = host launch -> event done.

    python tools/reserve_probe.py            (on the GPU box)
Modes: work on a plain stream | on a CU-masked stream (hiast_stream_create_reserved, 8 CUs = one per XCD left free)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hiast_amd import kernels as K  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    B, H, W, C = 8, 64, 128, 256
    x = torch.randn(B, H, W, C, device=dev).relu().half()
    w = torch.randn(C, C, 3, 3, device=dev) * 0.03
    wp = K.pack_conv_weight(w, K.FMT_FP16)
    probe_buf = torch.zeros(256, device=dev)
    probe_red = torch.zeros(1 << 16, device=dev)
    probe_out = torch.zeros((), device=dev)      # (sum over dim 0 of a 1-d tensor -> 0-d)
    probe_stream = torch.cuda.Stream(device=dev)
    kind = ["plain"]

    def probe_kernel():
        # "plain": one 256-thread block without LDS (co-resides with a tile-kernel block if registers allow);
        # "lds": a reduction (torch.sum: its blocks need shared memory) — like an RCCL kernel it cannot co-reside with a block
        # that holds all 160 KiB of a CU's LDS and has to wait for one to retire
        if kind[0] == "plain":
            probe_buf.add_(1.0)
        else:
            torch.sum(probe_red, dim=(0,), out=probe_out)

    def saturate(stream, n):
        with torch.cuda.stream(stream):
            y = x
            for _ in range(n):
                y = K.igemm_bn_act(y, wp, 1, None, None, True, 1, 2)
        return y

    def probe_alone(n=50):
        lat = []
        for _ in range(n):
            with torch.cuda.stream(probe_stream):
                e = torch.cuda.Event()
                t0 = time.perf_counter()
                probe_kernel()
                e.record()
                e.synchronize()
                lat.append(1e6 * (time.perf_counter() - t0))
            time.sleep(0.0002)
        return np.array(lat)

    def measure(stream, label):
        # warm up, then keep ~40 ms of tile kernels queued and probe beside them
        saturate(stream, 20)
        torch.cuda.synchronize()
        t_work0 = time.perf_counter()
        saturate(stream, 500)                               # ~35-40 ms of back-to-back 256-block launches
        t_enq = time.perf_counter() - t_work0
        lat = []
        done = torch.cuda.Event()
        done.record(stream)
        while not done.query():
            with torch.cuda.stream(probe_stream):
                e = torch.cuda.Event()
                t0 = time.perf_counter()
                probe_kernel()
                e.record()
                e.synchronize()
                lat.append(1e6 * (time.perf_counter() - t0))
            time.sleep(0.0002)
        torch.cuda.synchronize()
        total = time.perf_counter() - t_work0
        lat = np.array(lat) if lat else np.array([float("nan")])
        print("%-44s probes %3d | latency us: median %7.1f  p90 %7.1f  max %7.1f | 500 launches enqueued in %.1f ms, done after %.1f ms"
              % (label, len(lat), np.median(lat), np.percentile(lat, 90), lat.max(), 1e3 * t_enq, 1e3 * total))

    print("device CUs (hiast_device_cus): %d" % K.device_cus())
    plain = torch.cuda.Stream(device=dev)
    masked = {n: K.reserved_stream(n) for n in (8, 16)}
    for k in ("plain", "lds"):
        kind[0] = k
        print("---- probe kernel: %s" % ("one block, no LDS" if k == "plain" else "a reduction whose blocks need LDS (as an RCCL kernel does)"))
        a = probe_alone()
        print("%-44s probes %3d | latency us: median %7.1f  p90 %7.1f  max %7.1f" % ("probe alone (idle chip)", len(a), np.median(a),
                                                                                     np.percentile(a, 90), a.max()))
        measure(plain, "work on a plain stream (256 CUs)")
        for n in (8, 16):
            measure(masked[n], "work on a CU-masked stream (%d reserved)" % n)
        measure(plain, "work on a plain stream again")


if __name__ == "__main__":
    main()
