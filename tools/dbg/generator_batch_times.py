"""Per-batch device and host times of PSEUDO_POLICY['IAS'](cfg).run(): for every batch the device time of begin() (H2D +
normalise + forward + pass 1, HIP events on the launch stream), the host time of begin(), of the histogram wait and of
select_and_save_confident_label().
    python tools/dbg/generator_batch_times.py [N=496] [batch=8] [workers=14]"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hiast_amd.utils.registry import register  # noqa
from hiast_amd.utils.registry.registries import PSEUDO_POLICY  # noqa
from hiast_amd.tools import synth_data  # noqa

N = int(sys.argv[1]) if len(sys.argv) > 1 else 496
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nw = int(sys.argv[3]) if len(sys.argv) > 3 else 14
root = tempfile.mkdtemp(prefix="hiast_gen_")
try:
    cfg = synth_data.synthetic_cfg(root, n_train=N, n_val=1, h=512, w=1024, procs=max(nw, 1))
    cfg.pseudo_policy.batch_size = bs
    cfg.dataset.num_workers = nw
    gen = PSEUDO_POLICY["IAS"](cfg)
    gen.engine.pass1(torch.zeros((bs, 512, 1024, 3), dtype=torch.uint8)); gen.engine.pass2(None); torch.cuda.synchronize()
    eng = gen.engine
    rec = {"ev": [], "begin": [], "hist": [], "save": [], "fetch": []}
    begin, hist_host, save, batches = eng.begin, eng.hist_host, gen.select_and_save_confident_label, gen._batches

    def t_begin(imgs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t = time.perf_counter()
        e0.record()
        st = begin(imgs)
        e1.record()
        rec["begin"].append(time.perf_counter() - t)
        rec["ev"].append((e0, e1))
        return st

    def t_hist(st, ar=None):
        t = time.perf_counter()
        h = hist_host(st, ar)
        rec["hist"].append(time.perf_counter() - t)
        return h

    def t_save(paths, state=None):
        t = time.perf_counter()
        r = save(paths, state)
        rec["save"].append(time.perf_counter() - t)
        return r

    def t_batches():
        it = batches()
        while True:
            t = time.perf_counter()
            try:
                b = next(it)
            except StopIteration:
                return
            rec["fetch"].append(time.perf_counter() - t)
            yield b
    eng.begin, eng.hist_host, gen.select_and_save_confident_label, gen._batches = t_begin, t_hist, t_save, t_batches
    t0 = time.time()
    gen.run()
    torch.cuda.synchronize()
    dt = time.time() - t0
    dev = np.array([a.elapsed_time(b) for a, b in rec["ev"]])
    print("run: %.2f s for %d images = %.1f images/s" % (dt, N, N / dt))
    for k in ("fetch", "begin", "hist", "save"):
        v = 1e3 * np.array(rec[k])
        print("host %-6s ms: total %7.1f | median %6.2f | first five %s | max %.1f" % (k, v.sum(), np.median(v), np.round(v[:5], 1), v.max()))
    print("device begin() ms (event to event on the launch stream): total %.1f | median %.2f | first five %s | max %.1f"
          % (dev.sum(), np.median(dev), np.round(dev[:5], 1), dev.max()))
finally:
    shutil.rmtree(root, ignore_errors=True)
