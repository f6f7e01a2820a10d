"""Per-batch global histograms of the IAS generator: 2 ranks (gloo, both on cuda:0, batch 2) vs one process (batch 4 as two
sub-batches of 2), each run twice.  Prints where they differ."""
import os
import socket
import sys
import tempfile

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, cfg_dict, save_dir, batch, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        os.environ["HIAST_EVAL_SPLIT"] = "2"
    torch.cuda.set_device(0)
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import PSEUDO_POLICY
    from hiast_amd.utils.default_config import CfgNode
    from hiast_amd.workflows import ias_math
    c = CfgNode(cfg_dict)
    c.pseudo_policy.batch_size = batch
    c.pseudo_policy.save_dir = save_dir
    hists = []
    upd = ias_math.ias_update

    def rec(hist, *a, **k):
        hists.append(np.array(hist).copy())
        return upd(hist, *a, **k)
    ias_math.ias_update = rec
    gen = PSEUDO_POLICY["IAS"](c)
    # exact checksums of the library stem's outputs and of the logits, per invocation
    sums = []
    bb = gen.engine.model.seg_model.backbone
    f0 = bb.conv1.forward

    def f(x):
        y = f0(x)
        sums.append(("stem", tuple(x.shape), int(y.contiguous().view(torch.int32).long().sum().item())))
        return y
    bb.conv1.forward = f
    m0 = gen.engine.model.forward

    def mf(x, lowres=False):
        o = m0(x, lowres=lowres)
        sums.append(("logits", tuple(x.shape), int(o["logits_lowres"].float().contiguous().view(torch.int32).long().sum().item())))
        return o
    gen.engine.model.forward = mf
    gen.run()
    np.save(out + ".sums.%d.npy" % rank, np.array([repr(v) for v in sums]))
    if rank == 0:
        np.save(out, np.stack(hists))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import MODEL
    from hiast_amd.tools import synth_data
    from make_golden import seeded_state_dict
    root = tempfile.mkdtemp(prefix="hiast_shc_")
    cfg = synth_data.synthetic_cfg(root, n_train=8, n_val=1, h=128, w=256)
    m = MODEL["SelfTrainingSegmentor"](cfg)
    sd = {"seg_model." + k: v for k, v in seeded_state_dict(m.seg_model, 777).items()}
    for i in range(4):
        sd["seg_model.aspp.conv2d_list.%d.weight" % i] = sd["seg_model.aspp.conv2d_list.%d.weight" % i] * 40.0
        sd["seg_model.aspp.conv2d_list.%d.bias" % i] = sd["seg_model.aspp.conv2d_list.%d.bias" % i] * 40.0
    ck = os.path.join(root, "w.pth")
    torch.save(sd, ck)
    del m
    cfg.pseudo_policy.resume_from = ck
    res = {}
    for name, world, batch in (("w2a", 2, 2), ("w1a", 1, 4), ("w2b", 2, 2), ("w1b", 1, 4)):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        out = os.path.join(root, name + ".npy")
        mp.spawn(worker, args=(world, port, cfg.to_dict(), os.path.join(root, name, "pseudo_labels"), batch, out), nprocs=world, join=True)
        res[name] = np.load(out)
        print(name, res[name].shape, "pixels per batch", res[name].reshape(res[name].shape[0], -1).sum(1), flush=True)
    for name, world in (("w2a", 2), ("w1a", 1)):
        for r in range(world):
            print(name, "rank", r, list(np.load(os.path.join(root, name + ".npy.sums.%d.npy" % r))))
    for a, b in (("w2a", "w2b"), ("w1a", "w1b"), ("w2a", "w1a")):
        d = res[a].astype(np.int64) - res[b].astype(np.int64)
        print("%s vs %s: differing bins per batch %s, |sum of differences| per batch %s" % (a, b, (d != 0).reshape(d.shape[0], -1).sum(1), np.abs(d).reshape(d.shape[0], -1).sum(1)))
