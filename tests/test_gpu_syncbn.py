"""SyncBN through the fused BN kernels: 2 ranks (gloo, both on cuda:0) each holding half of a batch must
reproduce single-process BatchNorm over the whole batch — forward, running statistics and input gradient
(the data-parallel semantic of the reference's convert_syncbn_model, utils/utils.py:103-105)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import synth
    from hiast_amd import functional as HF
    B, C, H, W = 4, 32, 12, 20
    x = synth.normal_f32(301, (B, C, H, W), 2.0) + 0.3
    r = synth.normal_f32(302, (B, C, H, W), 1.0)
    g = synth.normal_f32(303, (B, C, H, W), 1.0)
    half = slice(rank * B // world, (rank + 1) * B // world)
    bn = torch.nn.SyncBatchNorm(C).cuda().train()
    xd = torch.from_numpy(x[half]).cuda().requires_grad_(True)
    rd = torch.from_numpy(r[half]).cuda().requires_grad_(True)
    y = HF.bn_act(xd, bn, rd, True)
    y.backward(torch.from_numpy(g[half]).cuda())
    res = {"y": y.detach().cpu().numpy(), "gx": xd.grad.cpu().numpy(), "gr": rd.grad.cpu().numpy(),
           "rm": bn.running_mean.cpu().numpy(), "rv": bn.running_var.cpu().numpy()}
    # the channels-last bf16 flavour (mixed-precision training trunk): one all-reduce of [C,2] double sums per pass
    bn2 = torch.nn.SyncBatchNorm(C).cuda().train()
    cl = lambda a: torch.from_numpy(a).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
    xb, rb = cl(x[half]).requires_grad_(True), cl(r[half]).requires_grad_(True)
    yb = HF.bn_act(xb, bn2, rb, True)
    assert yb.dtype == torch.bfloat16 and yb.permute(0, 2, 3, 1).is_contiguous()
    yb.backward(cl(g[half]))
    res.update({"yb": yb.detach().float().cpu().numpy(), "gxb": xb.grad.float().cpu().numpy(),
                "grb": rb.grad.float().cpu().numpy(), "rmb": bn2.running_mean.cpu().numpy(),
                "rvb": bn2.running_var.cpu().numpy()})
    np.savez(out % rank, **res)
    dist.barrier()
    dist.destroy_process_group()


def test_syncbn_two_ranks_equal_full_batch(tmp_path):
    import synth
    out = str(tmp_path / "r%d.npz")
    mp.spawn(_worker, args=(2, _port(), out), nprocs=2, join=True)
    B, C, H, W = 4, 32, 12, 20
    x = torch.from_numpy(synth.normal_f32(301, (B, C, H, W), 2.0) + 0.3).double().requires_grad_(True)
    r = torch.from_numpy(synth.normal_f32(302, (B, C, H, W), 1.0)).double().requires_grad_(True)
    g = torch.from_numpy(synth.normal_f32(303, (B, C, H, W), 1.0)).double()
    bn = torch.nn.BatchNorm2d(C).double().train()
    y = torch.relu(bn(x) + r)
    y.backward(g)
    parts = [np.load(out % k) for k in range(2)]
    got_y = np.concatenate([p["y"] for p in parts])
    got_gx = np.concatenate([p["gx"] for p in parts])
    got_gr = np.concatenate([p["gr"] for p in parts])
    assert np.allclose(got_y, y.detach().numpy(), rtol=1e-5, atol=1e-5)
    assert np.allclose(got_gr, r.grad.numpy(), rtol=1e-5, atol=1e-5)
    assert np.abs(got_gx - x.grad.numpy()).max() <= 1e-4 * np.abs(x.grad.numpy()).max()
    for p in parts:       # both ranks hold the GLOBAL running statistics
        assert np.allclose(p["rm"], bn.running_mean.numpy(), rtol=1e-5, atol=1e-6)
        assert np.allclose(p["rv"], bn.running_var.numpy(), rtol=1e-5, atol=1e-6)
    # channels-last bf16 flavour: same semantics on the bf16-rounded inputs, outputs rounded to bf16
    bf = lambda a: torch.from_numpy(a).bfloat16().double()
    xq = bf(synth.normal_f32(301, (B, C, H, W), 2.0) + 0.3).requires_grad_(True)
    rq = bf(synth.normal_f32(302, (B, C, H, W), 1.0)).requires_grad_(True)
    bnq = torch.nn.BatchNorm2d(C).double().train()
    yq = torch.relu(bnq(xq) + rq)
    yq.backward(bf(synth.normal_f32(303, (B, C, H, W), 1.0)))
    tol = lambda ref: 2.0 ** -7 * np.abs(ref) + 2e-3 * np.abs(ref).max()
    for key, ref in (("yb", yq.detach().numpy()), ("gxb", xq.grad.numpy()), ("grb", rq.grad.numpy())):
        got = np.concatenate([p[key] for p in parts])
        assert (np.abs(got - ref) <= tol(ref)).all(), key
    for p in parts:
        assert np.allclose(p["rmb"], bnq.running_mean.numpy(), rtol=1e-4, atol=1e-5)
        assert np.allclose(p["rvb"], bnq.running_var.numpy(), rtol=1e-4, atol=1e-5)
