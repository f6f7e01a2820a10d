"""validate.py's workflow on synthetic frames: VALIDATOR with flip TTA over N synthetic val images -> images/s end to end
(PNG decode in DataLoader workers -> H2D -> forwards -> fused TTA kernel -> IoU histogram).
    python tools/run_validator_synth.py [N=40] [batch=1] [workers=8] [H=1024] [W=2048]"""
import os, sys, time, tempfile, shutil
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from hiast_amd.utils.registry import register  # noqa
from hiast_amd.tools import synth_data
from hiast_amd.workflows.validator import Validator

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nw = int(sys.argv[3]) if len(sys.argv) > 3 else 8
Hh = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
Ww = int(sys.argv[5]) if len(sys.argv) > 5 else 2048
root = tempfile.mkdtemp(prefix="hiast_val_")
try:
    cfg = synth_data.synthetic_cfg(root, n_train=1, n_val=N, h=Hh, w=Ww, procs=max(nw, 1))
    cfg.validate.batch_size = bs
    cfg.validate.is_flip = True
    cfg.validate.resize_sizes = [[Hh, Ww]]
    cfg.dataset.num_workers = nw
    v = Validator(cfg)
    v.run()                         # warm-up epoch (kernel load, worker start)
    torch.cuda.synchronize()
    t0 = time.time()
    v.run()
    torch.cuda.synchronize()
    dt = time.time() - t0
    print("Validator: %d frames (%dx%d, bs %d, flip TTA, %d workers) in %.2fs = %.1f frames/s; mIoU %.4f" % (N, Ww, Hh, bs, nw, dt, N / dt, v.miou))
finally:
    shutil.rmtree(root, ignore_errors=True)
