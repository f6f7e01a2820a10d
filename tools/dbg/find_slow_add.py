"""Which aten::add of the backward pass takes the strided (non-vectorised) path: shapes + strides of every add."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    torch.cuda.set_device(0)
    cfg = bench.make_cfg(1, "ConsistencySelfTrainingTrainer")
    hp = bench.HotPath(cfg, torch.device("cuda", 0), 0, 1, 8)
    for _ in range(2):
        hp.step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
        hp.step()
        torch.cuda.synchronize()
    rows = []
    for ev in prof.events():
        if ev.name in ("aten::add", "aten::add_", "aten::copy_", "aten::contiguous", "aten::clone", "aten::to", "aten::_to_copy") and ev.device_time_total > 20:
            rows.append((ev.device_time_total, ev.name, str(ev.input_shapes)[:150]))
    rows.sort(reverse=True)
    for r in rows[:25]:
        print("%9.1f us  %-18s %s" % r)


if __name__ == "__main__":
    main()
