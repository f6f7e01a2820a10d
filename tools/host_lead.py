"""How far the host runs AHEAD of the device at points of one bench step (unprofiled): for each mark, the host time
at which the mark was enqueued vs the device time at which the stream reached it.  lead ~ 0 => the device was waiting
for the host there (an idle gap follows); a large lead => the launch queue was full.
    python tools/host_lead.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    cfg = bench.make_cfg(1, "ConsistencySelfTrainingTrainer")
    hp = bench.HotPath(cfg, dev, 0, 1, 8)
    from hiast_amd import functional as HF
    for _ in range(3):
        hp.step()
    torch.cuda.synchronize()
    marks = []

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((name, time.perf_counter(), e))

    # re-implementation of HotPath.step with marks
    def step():
        mark("step_begin")
        mp, am = hp.plabel_begin()
        mark("plabel_begin_enqueued")
        out, teacher_lr = hp.train_forward()
        mark("train_forwards_enqueued")
        plbl = hp.plabel_finish(mp, am)
        mark("hist_wait+thresholds+pass2")
        losses = hp.model.module.compute_loss_lowres(out["logits_lowres"], plbl, out["size"], teacher_lr)
        g_loss = sum(torch.mean(v) for v in losses.values())
        hp.opt.zero_grad(set_to_none=True)
        mark("loss_enqueued")
        g_loss.backward()
        mark("backward_enqueued")
        HF.wgrad_stream_join()
        hp.opt.step()
        mark("adam_enqueued")
        hp.ema_updater(hp.ema, hp.model, cfg.cst_training.ema_model.gamma)
        mark("ema_enqueued")
        for s in hp.sched:
            s.step()
        mark("step_end")

    base = torch.cuda.Event(enable_timing=True)
    base.record()
    base.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    rows = [(n, (t - t0) * 1e3, base.elapsed_time(e)) for n, t, e in marks]
    print("%-30s %10s %10s %8s %10s" % ("mark", "host ms", "device ms", "lead", "host d"))
    prev_h = 0.0
    for n, h, d in rows[-2 * 9:]:
        print("%-30s %10.2f %10.2f %8.2f %10.2f" % (n, h, d, d - h, h - prev_h))
        prev_h = h
    print("ms/step: %.2f" % ((rows[-1][2] - rows[0][2]) / steps))


if __name__ == "__main__":
    main()
