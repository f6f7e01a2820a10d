"""BASELINE configs[1] as the real workflow: PSEUDO_POLICY['IAS'](cfg).run() over N synthetic 1024x512 target
images (PNG decode in DataLoader workers -> H2D -> fp32 forward -> pass 1 -> thresholds -> pass 2 -> D2H ->
PNG encode on a thread pool).  Prints end-to-end images/s (PCIe- and IO-inclusive).
    python tools/run_generator_synth.py [N=64] [batch=8] [workers=8] [H=512] [W=1024]"""
import os, sys, time, tempfile, shutil
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from hiast_amd.utils.registry import register  # noqa
from hiast_amd.utils.registry.registries import PSEUDO_POLICY
from hiast_amd.tools import synth_data

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nw = int(sys.argv[3]) if len(sys.argv) > 3 else 8
Hh = int(sys.argv[4]) if len(sys.argv) > 4 else 512
Ww = int(sys.argv[5]) if len(sys.argv) > 5 else 1024
root = tempfile.mkdtemp(prefix="hiast_gen_")
try:
    t0 = time.time()
    cfg = synth_data.synthetic_cfg(root, n_train=N, n_val=1, h=Hh, w=Ww, procs=max(nw, 1))
    print("wrote %d synthetic images in %.1fs" % (N, time.time() - t0), flush=True)
    cfg.pseudo_policy.batch_size = bs
    cfg.dataset.num_workers = nw
    t0 = time.time()
    gen = PSEUDO_POLICY["IAS"](cfg)         # starts the DataLoader workers, then loads the model onto the device
    t_init = time.time() - t0
    # warm-up: one synthetic batch through the engine (kernel load / library algorithm search), not counted
    gen.engine.pass1(torch.zeros((bs, Hh, Ww, 3), dtype=torch.uint8)); gen.engine.pass2(None); torch.cuda.synchronize()
    t0 = time.time()
    gen.run()
    torch.cuda.synchronize()
    dt = time.time() - t0
    print("IAS generator: constructor (worker start + model to the device) %.2fs; run(): %d images (%dx%d, bs %d, %d "
          "workers) in %.2fs = %.1f images/s end to end (PNG decode -> ... -> PNG files written)" % (t_init, N, Ww, Hh, bs, nw, dt, N / dt))
finally:
    shutil.rmtree(root, ignore_errors=True)
