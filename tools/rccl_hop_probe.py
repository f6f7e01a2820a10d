"""What does ONE SyncBN exchange cost on the stream before any link time?  (one rank, torch's RCCL backend)

A SyncBN layer of the N > 1 step is  producer kernel -> [C,2] all-reduce -> consumer kernel  on the main stream, 208 times per
step.  This probe times that chain with HIP events on ONE rank of backend 'nccl' (a one-rank all-reduce moves nothing, so
what is left is the backend's own cost on the stream: the call, the stream hand-offs, the work handle):

    chain of 200 x [k; k]                                  (no exchange)
    chain of 200 x [k; all_reduce(sync);  k]               (the forward exchanges, and the backward ones with HIAST_NO_ASYNC_STAT=1)
    chain of 200 x [k; all_reduce(async); k2 on main; wait; k]   (the early backward exchange: k2 stands for the weight gradient)

Two regimes: k = a 4 us elementwise kernel on the [1024, 2] double tensor (the chain is then bound by the HOST: what the call
costs the enqueueing thread) and k = a ~30 us streaming kernel (the host runs ahead: what the exchange costs on the STREAM).
    python tools/rccl_hop_probe.py   (on the GPU box)"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29543")
    os.environ["HIAST_DIST_REHEARSAL"] = "1"
    torch.cuda.set_device(0)
    from hiast_amd.utils import comm
    comm.init_process_group("nccl", rank=0, world_size=1)
    dev = torch.device("cuda", 0)
    t = torch.zeros(1024, 2, dtype=torch.float64, device=dev)
    u = torch.zeros(1 << 20, device=dev)
    N = 200

    v = torch.zeros(16 << 20, device=dev)
    heavy = [False]

    def k():
        if heavy[0]:
            v.add_(1.0)         # 64 MB read + written: ~30 us
        else:
            t.add_(1.0)

    def k2():
        u.mul_(1.0001)          # ~10 us: independent work on the main stream

    def plain():
        for _ in range(N):
            k(); k()

    def plain2():
        for _ in range(N):
            k(); k2(); k()

    def sync():
        for _ in range(N):
            k(); comm.all_reduce(t, "stat"); k()

    def asyn():
        for _ in range(N):
            k(); w = comm.all_reduce(t, "stat", async_op=True); k2(); comm.wait(w, "stat"); k()

    def default_group():
        for _ in range(N):
            k(); dist.all_reduce(t); k()

    def run(f):
        f()
        torch.cuda.synchronize()
        best = (1e9, 1e9)
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            h0 = time.perf_counter()
            e0.record()
            f()
            e1.record()
            h1 = time.perf_counter()
            torch.cuda.synchronize()
            best = min(best, (e0.elapsed_time(e1) * 1e3 / N, (h1 - h0) * 1e6 / N))
        return best

    for hv in (False, True):
        heavy[0] = hv
        base, base2 = run(plain), run(plain2)
        print("k = %s; per iteration, best of 5 (stream time by HIP events | host time to enqueue), %d iterations per chain"
              % ("~30 us streaming kernel (host ahead)" if hv else "4 us kernel (host-bound chain)", N))
        print("  [k; k]                                   %6.1f us | %6.1f us" % base)
        print("  [k; k2; k]                               %6.1f us | %6.1f us" % base2)
        for name, f, ref in (("[k; all_reduce sync (stat group); k]", sync, base),
                             ("[k; all_reduce sync (default group); k]", default_group, base),
                             ("[k; all_reduce async; k2; wait; k]", asyn, base2)):
            r = run(f)
            print("  %-40s %6.1f us | %6.1f us   -> the exchange adds %5.1f us on the stream, %5.1f us on the host"
                  % (name, r[0], r[1], r[0] - ref[0], r[1] - ref[1]))
    dist.barrier(device_ids=[0])
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
