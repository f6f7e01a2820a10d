"""cProfile of the main thread of PSEUDO_POLICY['IAS'](cfg).run() on synthetic 1024x512 images:
    python tools/dbg/profile_generator_host.py [N=128] [batch=2] [workers=8]"""
import cProfile
import os
import pstats
import shutil
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hiast_amd.utils.registry import register  # noqa
from hiast_amd.utils.registry.registries import PSEUDO_POLICY  # noqa
from hiast_amd.tools import synth_data  # noqa

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nw = int(sys.argv[3]) if len(sys.argv) > 3 else 8
root = tempfile.mkdtemp(prefix="hiast_gen_")
try:
    cfg = synth_data.synthetic_cfg(root, n_train=N, n_val=1, h=512, w=1024, procs=max(nw, 1))
    cfg.pseudo_policy.batch_size = bs
    cfg.dataset.num_workers = nw
    gen = PSEUDO_POLICY["IAS"](cfg)
    gen.engine.pass1(torch.zeros((bs, 512, 1024, 3), dtype=torch.uint8)); gen.engine.pass2(None); torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.time()
    pr.enable()
    gen.run()
    torch.cuda.synchronize()
    pr.disable()
    print("run: %.2f s for %d images = %.1f images/s" % (time.time() - t0, N, N / (time.time() - t0)))
    pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
finally:
    shutil.rmtree(root, ignore_errors=True)
