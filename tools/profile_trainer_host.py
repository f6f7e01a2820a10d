"""cProfile of the HOST side of the real trainer's iteration (TRAINER['ConsistencySelfTrainingTrainer'].step minus the
DataLoader: fixed device-resident batch): python tools/profile_trainer_host.py"""
import cProfile
import os
import pstats
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402


def main():
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.workflows.trainer.consistency_self_training_trainer import ConsistencySelfTrainingTrainer

    class StepOnly(ConsistencySelfTrainingTrainer):
        def assert_cfg(self):
            pass

        def build_train_data_reader(self):
            pass

        def build_val_data_reader(self):
            self.v_loader = None

        def train(self):
            return self.train_on(self.weak, self.strong, self.plbl)

    cfg = bench.make_cfg(1, "ConsistencySelfTrainingTrainer")
    cfg.work_dir = tempfile.mkdtemp()
    cfg.train.iter_report = 10 ** 6
    cfg.train.iter_val = 10 ** 6
    tr = StepOnly(cfg, 0)
    g = torch.Generator().manual_seed(1)
    tr.weak = torch.randn(8, 3, 512, 1024, generator=g).cuda()
    tr.strong = (tr.weak * 1.05 + 0.02).contiguous()
    tr.plbl = torch.randint(0, 19, (8, 512, 1024), dtype=torch.uint8).cuda()
    tr.model_recorder.reset_time_and_losses()
    for it in range(1, 5):
        tr.step(it)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for it in range(5, 13):
        tr.step(it)
    th = (time.perf_counter() - t0) / 8
    torch.cuda.synchronize()
    tt = (time.perf_counter() - t0) / 8
    print("trainer step: host loop %.1f ms/iter, with final drain %.1f ms/iter" % (th * 1e3, tt * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for it in range(13, 17):
        tr.step(it)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
    st.sort_stats("cumulative").print_stats(40)


if __name__ == "__main__":
    main()
