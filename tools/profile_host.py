"""cProfile of the HOST side of bench.py's step (where the ~40 ms of Python per step go): python tools/profile_host.py"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8       # python tools/profile_host.py [images per step]
    torch.cuda.set_device(0)
    cfg = bench.make_cfg(1, "ConsistencySelfTrainingTrainer")
    hp = bench.HotPath(cfg, torch.device("cuda", 0), 0, 1, batch)
    for _ in range(3):
        hp.step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(4):
        hp.step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(60)
    st.sort_stats("cumulative").print_stats(70)


if __name__ == "__main__":
    main()
