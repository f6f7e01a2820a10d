"""BASELINE configs[2] as the real workflow, end to end THROUGH THE DATALOADER: TRAINER['ConsistencySelfTrainingTrainer']
on N synthetic Cityscapes-size (2048x1024) target images whose pseudo-labels were written by PSEUDO_POLICY['IAS'] first
(PNG decode + CopyPaste + flip / random-sized crop / resize to 1024x512 + two colour views in DataLoader worker
processes -> H2D -> teacher forward, student forward/backward, Adam, EMA).  Prints images/s including the host data path
next to the compute-only number bench.py reports.
    python tools/run_trainer_synth.py [N=32] [batch=8] [workers=14] [iters=24] [native_h=1024] [native_w=2048]"""
import os
import shutil
import sys
import tempfile
import time

import torch

LOG = os.environ.get("HIAST_LOG")        # progress also goes to this file (gpurun watches gpurun_out/ for liveness)
_print = print


def print(*a, **k):     # noqa: A001
    _print(*a, **k)
    if LOG:
        with open(LOG, "a") as f:
            _print(*a, file=f)


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hiast_amd.utils.registry import register  # noqa: E402,F401
from hiast_amd.utils.registry.registries import MODEL, PSEUDO_POLICY, TRAINER  # noqa: E402
from hiast_amd.tools import synth_data  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nw = int(sys.argv[3]) if len(sys.argv) > 3 else 14
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 24
nh = int(sys.argv[5]) if len(sys.argv) > 5 else 1024
nwid = int(sys.argv[6]) if len(sys.argv) > 6 else 2048
root = tempfile.mkdtemp(prefix="hiast_train_")
try:
    t0 = time.time()
    cfg = synth_data.synthetic_cfg(root, n_train=N, n_val=2, h=nh, w=nwid, upscale=2, procs=min(nw, 14))
    print("wrote %d synthetic %dx%d images in %.1fs" % (N, nwid, nh, time.time() - t0), flush=True)
    torch.manual_seed(888)
    m = MODEL["SelfTrainingSegmentor"](cfg)
    ck = os.path.join(root, "warmup.pth")
    torch.save(m.state_dict(), ck)
    del m
    cfg.pseudo_policy.resume_from = ck
    cfg.pseudo_policy.batch_size = 4
    cfg.pseudo_policy.resize_size = [nh, nwid]
    cfg.dataset.num_workers = nw
    t0 = time.time()
    PSEUDO_POLICY["IAS"](cfg).run()
    torch.cuda.synchronize()
    print("IAS generator at %dx%d: %d images in %.1fs (first call, includes kernel load)" % (nwid, nh, N, time.time() - t0),
          flush=True)

    cfg.trainer = "ConsistencySelfTrainingTrainer"
    cfg.train.resume_from = ck
    cfg.train.gpu_num = 1
    cfg.train.batch_size = bs
    cfg.train.total_iter = 10 ** 6
    cfg.train.iter_report = 10 ** 6
    cfg.train.iter_val = 10 ** 6
    cfg.train.lr = 3e-6
    cfg.dataset.target.pseudo_dir = cfg.pseudo_policy.save_dir
    cfg.dataset.target.aug_type = ["MS", "CCA"]
    cfg.cst_training.is_enabled = True
    cfg.cst_training.cst_loss.weight = 0.5
    cfg.preprocessor.type = "CopyPaste"
    cfg.work_dir = os.path.join(root, "work")
    def measure(tag, warm=6):
        tr = TRAINER[cfg.trainer](cfg, 0)
        wait = [0.0]
        fetch = tr.next_target_batch

        def timed_fetch():                      # time the main process spends waiting for / receiving a batch
            a = time.time()
            b = fetch()
            wait[0] += time.time() - a
            return b
        tr.next_target_batch = timed_fetch
        # where the main process's time goes (HIAST_E2E_BREAKDOWN=1): wall time inside the parts of an iteration
        parts = {}
        if os.environ.get("HIAST_E2E_BREAKDOWN", "0") == "1":
            from hiast_amd.sseg.datasets import utils as du_

            def wrap(obj, name, key):
                f = getattr(obj, name)

                def g(*a, **k):
                    t = time.perf_counter()
                    r = f(*a, **k)
                    parts[key] = parts.get(key, 0.0) + time.perf_counter() - t
                    return r
                setattr(obj, name, g)
            wrap(du_, "to_device_batch", "to_device_batch")
            wrap(tr, "train_on", "train_on (teacher + student forward, loss)")
            wrap(tr, "update_model", "update_model (backward, Adam)")
            wrap(tr, "after_update", "after_update (EMA)")
            wrap(tr.model_recorder, "record_losses", "record_losses")
        for it in range(1, warm + 1):
            tr.step(it)
        torch.cuda.synchronize()
        if os.environ.get("HIAST_SYNC_DEBUG", "0") == "1":       # report every implicitly synchronising call of an iteration
            torch.cuda.set_sync_debug_mode("warn")
            tr.step(warm + 1)
            torch.cuda.set_sync_debug_mode("default")
            torch.cuda.synchronize()
        wait[0] = 0.0
        parts.clear()
        t0 = time.time()
        c0 = time.process_time()                # CPU time of the main process: a host loop that is BLOCKED on a full launch queue is
                                                # not busy (wall time alone cannot tell the two apart)
        idle = 0                                # iterations at whose end the device had already finished everything enqueued
        for it in range(warm + 1, warm + iters + 1):
            tr.step(it)
            ev = torch.cuda.Event()
            ev.record()
            idle += int(ev.query())
        t_host = (time.time() - t0) / iters     # host time per iteration (enqueue + data), before the final drain
        t_cpu = (time.process_time() - c0) / iters
        print("  device already idle at the end of %d of %d iterations (host-bound iterations)" % (idle, iters), flush=True)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / iters
        print("  host loop %.1f ms/iter wall (%.1f ms of CPU time in the main process) of which %.1f ms in next_target_batch()"
              % (t_host * 1e3, t_cpu * 1e3, wait[0] / iters * 1e3), flush=True)
        if parts:
            print("  breakdown (ms/iter): " + ", ".join("%s %.1f" % (k, v / iters * 1e3) for k, v in parts.items()), flush=True)
        del tr.next_target_batch                # (the instance attribute; the class method is back)
        print("ConsistencySelfTrainingTrainer end to end, %s (DataLoader, %d workers, CopyPaste + MS + CCA, bs %d): "
              "%.1f ms/iter = %.1f images/s" % (tag, nw, bs, dt * 1e3, bs / dt), flush=True)
        # host data path alone: how fast can the workers deliver batches?
        t0 = time.time()
        for _ in range(iters):
            tr.next_target_batch()
        dl = (time.time() - t0) / iters
        print("DataLoader alone, %s: %.1f ms/batch = %.1f images/s with %d workers" % (tag, dl * 1e3, bs / dl, nw), flush=True)
        # the same trainer on ONE batch, workers idle: (a) host-resident batch (H2D + normalise + step every iteration),
        # (b) device-resident batch (the step alone) — what the device needs for an iteration on this data
        if os.environ.get("HIAST_E2E_FIXED", "1") == "1":
            from hiast_amd.sseg.datasets import utils as du_
            b = tr.next_target_batch()
            time.sleep(3.0)                     # the workers fill their prefetch queues and go idle
            tr.next_target_batch = lambda: b
            for label, resident in (("host-resident", False), ("device-resident", True)):
                orig = du_.to_device_batch
                if resident:
                    cache = {}

                    def cached(imgs, lbl, dev, _o=orig, _c=cache):
                        if "v" not in _c:
                            _c["v"] = _o(imgs, lbl, dev)
                        return _c["v"]
                    du_.to_device_batch = cached
                try:
                    for it in range(3):
                        tr.step(10 ** 5 + it)
                    torch.cuda.synchronize()
                    t0 = time.time()
                    for it in range(iters):
                        tr.step(10 ** 5 + 10 + it)
                    th = (time.time() - t0) / iters
                    torch.cuda.synchronize()
                    dt = (time.time() - t0) / iters
                finally:
                    du_.to_device_batch = orig
                print("  one %s batch, workers idle: %.1f ms/iter (host loop %.1f)" % (label, dt * 1e3, th * 1e3), flush=True)
            del tr.next_target_batch
        tr.t_iter = tr.t_loader = None          # stop this trainer's worker processes before the next measurement
        del tr
        import gc
        gc.collect()
        torch.cuda.empty_cache()

    measure("PNG decode per sample (reference behaviour)")
    # decoded-image cache (cfg.dataset.decoded_cache_dir): every image / pseudo-label is decoded once, then re-read as raw bytes
    dcache = tempfile.mkdtemp(prefix="hiast_dcache_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    cfg.dataset.decoded_cache_dir = dcache
    cfg.dataset.decoded_cache_gb = 8.0
    try:
        measure("decoded-image cache (the warm-up iterations fill it)", warm=max(6, 2 * N // bs))
        used = sum(os.path.getsize(os.path.join(dcache, f)) for f in os.listdir(dcache))
        print("decoded cache: %d files, %.0f MB" % (len(os.listdir(dcache)), used / 1e6), flush=True)
    finally:
        shutil.rmtree(dcache, ignore_errors=True)
finally:
    shutil.rmtree(root, ignore_errors=True)
