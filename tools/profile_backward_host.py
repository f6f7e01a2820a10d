"""Host time of the BACKWARD pass per autograd node (the autograd engine runs the Python backward functions on its own
thread, which cProfile of the main thread does not see): python tools/profile_backward_host.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    torch.cuda.set_device(0)
    cfg = bench.make_cfg(1, "ConsistencySelfTrainingTrainer")
    hp = bench.HotPath(cfg, torch.device("cuda", 0), 0, 1, 8)
    for _ in range(3):
        hp.step()
    torch.cuda.synchronize()
    from hiast_amd import functional as HF
    mp, am = hp.plabel_begin()
    out, teacher = hp.train_forward()
    plbl = hp.plabel_finish(mp, am)
    losses = hp.model.module.compute_loss_lowres(out["logits_lowres"], plbl, out["size"], teacher)
    g_loss = sum(torch.mean(v) for v in losses.values())
    hp.opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    with torch.autograd.profiler.profile(use_device=None) as prof:
        g_loss.backward()
    HF.wgrad_stream_join()
    torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=60))


if __name__ == "__main__":
    main()
