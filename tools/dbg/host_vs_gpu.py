"""host enqueue time vs GPU time of the bench step's two phases"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
args = type("A", (), {})()
cfg = bench.make_cfg(1, "ConsistencySelfTrainingTrainer")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
hp = bench.HotPath(cfg, dev, 0, 1, 8)
for _ in range(3): hp.step()
torch.cuda.synchronize()
for it in range(4):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    torch.cuda.synchronize()
    t0 = time.perf_counter(); e[0].record()
    plbl = hp.plabel_pass()
    t1 = time.perf_counter(); e[1].record()
    hp.train_step(plbl)
    t2 = time.perf_counter(); e[2].record()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print("plabel: host %.1f ms gpu %.1f ms | train: host-enqueue %.1f ms gpu %.1f ms | wall %.1f ms" % (
        (t1 - t0) * 1e3, e[0].elapsed_time(e[1]), (t2 - t1) * 1e3, e[1].elapsed_time(e[2]), (t3 - t0) * 1e3), flush=True)
