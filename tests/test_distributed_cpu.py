"""world_size-2 gloo tests (CPU) of the N>1 paths: the sharded pseudo-label generator (host logic, with
the oracle-backed engine standing in for the HIP kernels) must equal the single-process run at
batch_size = world * local_batch on the same image order, and the packed loss all-reduce of the
recorder."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _gen_worker(rank, world, port, root, out, policy="IAS", pipelined=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import PSEUDO_POLICY
    from hiast_amd.tools import synth_data
    from test_host_cpu import OracleEngine, PipelinedOracleEngine
    if pipelined:       # the begin / hist_host / finish interface: the generators run their software-pipelined loop
        OracleEngine = PipelinedOracleEngine
    h, w, C = 32, 64, 19
    c = synth_data.synthetic_cfg(root, n_train=7, n_val=1, h=h, w=w) if rank == 0 else None
    dist.barrier()
    objs = [c.to_dict() if rank == 0 else None]
    dist.broadcast_object_list(objs, src=0)
    from hiast_amd.utils.default_config import CfgNode
    c = CfgNode(objs[0])
    c.pseudo_policy.batch_size = 2
    c.pseudo_policy.type = policy
    c.pseudo_policy.cbst.sample_interval = 3
    c.pseudo_policy.save_dir = os.path.join(root, "pseudo_w%d" % world, "pseudo_labels")
    gen = PSEUDO_POLICY[policy](c, engine=OracleEngine(C, h, w))
    gen.run()
    if rank == 0:
        np.save(out, gen.class_threshold)
    dist.barrier()
    # labels exist now: a second run() must return on EVERY rank (rank 0 sees the files, the others take its word) —
    # a rank that went on alone would block in the histogram all-reduce
    n_before = len(gen.sample_stats)
    gen.run()
    assert len(gen.sample_stats) == n_before
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("pipelined", [False, True])
def test_sharded_generator_equals_single_process(tmp_path, pipelined):
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import PSEUDO_POLICY
    from hiast_amd.utils.default_config import CfgNode
    from hiast_amd.tools import synth_data
    from test_host_cpu import OracleEngine
    from PIL import Image
    root = str(tmp_path)
    out = os.path.join(root, "thr_w2.npy")
    mp.spawn(_gen_worker, args=(2, _free_port(), root, out, "IAS", pipelined), nprocs=2, join=True)
    # single process, batch 4 (= world 2 x local 2), same seeded order
    h, w, C = 32, 64, 19
    c = synth_data.synthetic_cfg(root + "/again", n_train=7, n_val=1, h=h, w=w)
    # reuse the SAME images as the 2-rank run
    first = json.load(open(os.path.join(root, "data", "cityscapes_train.json")))
    c.dataset.target.json_path = os.path.join(root, "data", "cityscapes_train.json")
    c.dataset.target.image_dir = os.path.join(root, "data", "cityscapes")
    assert len(first) == 7
    c.pseudo_policy.batch_size = 4
    c.pseudo_policy.save_dir = os.path.join(root, "pseudo_w1", "pseudo_labels")
    gen = PSEUDO_POLICY["IAS"](c, engine=OracleEngine(C, h, w))
    gen.run()
    thr2 = np.load(out)
    assert np.array_equal(thr2.view(np.uint64), gen.class_threshold.view(np.uint64))
    d1, d2 = os.path.join(root, "pseudo_w1"), os.path.join(root, "pseudo_w2")
    for f in ("statics_class.npy", "class_mean_probabilities.npy", "class_threshold.npy"):
        a, b = np.load(os.path.join(d1, f)), np.load(os.path.join(d2, f))
        assert np.array_equal(a, b), f
    names = sorted(os.listdir(os.path.join(d1, "pseudo_labels")))
    assert names == sorted(os.listdir(os.path.join(d2, "pseudo_labels"))) and len(names) == 7
    for n in names:
        assert np.array_equal(np.array(Image.open(os.path.join(d1, "pseudo_labels", n))),
                              np.array(Image.open(os.path.join(d2, "pseudo_labels", n)))), n
    s1 = json.load(open(os.path.join(d1, "sample_class_stats.json")))
    s2 = json.load(open(os.path.join(d2, "sample_class_stats.json")))
    key = lambda s: s["file"]
    assert sorted(s1, key=key) == sorted(s2, key=key)
    w1 = json.load(open(os.path.join(d1, "samples_with_class.json")))
    w2 = json.load(open(os.path.join(d2, "samples_with_class.json")))
    assert {k: sorted(v) for k, v in w1.items()} == {k: sorted(v) for k, v in w2.items()}


def test_sharded_cbst_equals_single_process(tmp_path):
    """CBST's strided confidence sample ranks the pixels of a GLOBAL batch across the ranks: 2 ranks x batch 2 give the
    thresholds (and label maps) of the single process at batch 4, bit for bit, and both equal the reference's
    list formulation (oracle) driven in the same order"""
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import PSEUDO_POLICY
    from hiast_amd.tools import synth_data
    from oracle import ias_ref
    from test_host_cpu import OracleEngine
    root = str(tmp_path)
    out = os.path.join(root, "thr_w2.npy")
    mp.spawn(_gen_worker, args=(2, _free_port(), root, out, "CBST"), nprocs=2, join=True)
    h, w, C = 32, 64, 19
    c = synth_data.synthetic_cfg(root + "/again", n_train=7, n_val=1, h=h, w=w)
    c.dataset.target.json_path = os.path.join(root, "data", "cityscapes_train.json")
    c.dataset.target.image_dir = os.path.join(root, "data", "cityscapes")
    c.pseudo_policy.type = "CBST"
    c.pseudo_policy.cbst.sample_interval = 3
    c.pseudo_policy.batch_size = 4
    c.pseudo_policy.save_dir = os.path.join(root, "pseudo_w1", "pseudo_labels")
    gen = PSEUDO_POLICY["CBST"](c, engine=OracleEngine(C, h, w))
    gen.run()
    thr2 = np.load(out)
    assert np.array_equal(thr2.view(np.uint64), gen.class_threshold.view(np.uint64), equal_nan=False) or \
        np.array_equal(np.nan_to_num(thr2, nan=-1.0), np.nan_to_num(gen.class_threshold, nan=-1.0))
    # the reference's formulation on the same batches
    eng = OracleEngine(C, h, w)
    batches = []
    for data in gen.t_loader:
        eng.pass1(data["images"])
        batches.append((eng.mp, eng.am.astype(np.int64)))
    want = ias_ref.cbst_threshold(batches, C, c.pseudo_policy.cbst.p, 3, as_float64=True)
    assert np.array_equal(np.nan_to_num(gen.class_threshold, nan=-1.0), np.nan_to_num(want, nan=-1.0))
    for f in ("statics_class.npy", "class_mean_probabilities.npy"):
        a, b = np.load(os.path.join(root, "pseudo_w1", f)), np.load(os.path.join(root, "pseudo_w2", f))
        assert np.array_equal(a, b), f


def test_sharded_batch_sampler_partitions():
    from hiast_amd.workflows.pseudo_label_generator import ShardedBatchSampler
    for n, bs, world in [(7, 2, 2), (10, 3, 4), (4, 2, 8), (0, 2, 2)]:
        parts = [list(ShardedBatchSampler(n, bs, r, world, True, 5)) for r in range(world)]
        steps = {len(p) for p in parts}
        assert len(steps) == 1                         # every rank takes the same number of steps
        seen = [i for p in parts for b in p for i in b]
        assert sorted(seen) == list(range(n))
        single = [i for b in ShardedBatchSampler(n, bs * world, 0, 1, True, 5) for i in b]
        inter = [i for t in range(len(parts[0])) for r in range(world) for i in parts[r][t]]
        assert inter == single                         # rank-major order inside a global batch


def _rec_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hiast_amd.utils.default_config import get_default_cfg
    from hiast_amd.utils.result_recorder import ResultRecorder
    rec = ResultRecorder(get_default_cfg(), rank, None, None, "model")
    for it in range(3):
        rec.record_losses({"a": torch.tensor(1.0 + rank + it), "b": torch.tensor(10.0 * (rank + 1))})
    vals = rec.report_losses(3)
    if rank == 0:
        json.dump(vals, open(out, "w"))
    dist.destroy_process_group()


def test_recorder_packs_losses_into_one_allreduce(tmp_path):
    out = str(tmp_path / "v.json")
    mp.spawn(_rec_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    v = json.load(open(out))
    assert abs(v["a"] - 2.5) < 1e-6 and abs(v["b"] - 15.0) < 1e-6


def _grad_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hiast_amd.utils import utils
    torch.manual_seed(3)
    params = [torch.nn.Parameter(torch.zeros(s)) for s in [(5, 3), (7,), (2, 2, 3), (1,), (300,)]]
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float((rank + 1) * (i + 1))) + torch.arange(p.numel()).view(p.shape)
    params.append(torch.nn.Parameter(torch.zeros(4)))           # a parameter without gradient is skipped
    utils.all_reduce_grads(params, world, bucket_bytes=64)      # tiny buckets: several flushes
    if rank == 0:
        torch.save([p.grad for p in params], out)
    dist.destroy_process_group()


def test_manual_gradient_allreduce_averages_over_ranks(tmp_path):
    """the adversarial warm-up trainer's own gradient exchange (two backward passes per forward, no DDP wrapper)"""
    out = str(tmp_path / "g.pt")
    mp.spawn(_grad_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    grads = torch.load(out)
    assert grads[-1] is None
    for i, g in enumerate(grads[:-1]):
        want = torch.full_like(g, 1.5 * (i + 1)) + torch.arange(g.numel()).view(g.shape)
        assert torch.equal(g, want)


def _comm_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hiast_amd import functional as HF
    from hiast_amd.utils import comm
    comm.setup()
    sg, ag = comm.stat_group(), comm.aux_group()
    default = dist.distributed_c10d._get_default_group()
    ok = sg is not None and ag is not None and sg is not ag and sg is not default and ag is not default
    ok = ok and comm.stat_group() is sg and comm.aux_group() is ag           # created once
    # the three communicators carry independent sequences: each rank issues them in a different interleaving per group
    # kind, as DDP buckets / SyncBN sums / the histogram do in a step, and every sum must still be the global one
    g = torch.full((1 << 16,), float(rank + 1), dtype=torch.float32)        # a "gradient bucket" on the default group
    s = torch.tensor([[1.0 + rank, 2.0 * rank]] * 7, dtype=torch.float64)   # SyncBN [C,2] double sums
    h = torch.arange(19 * 11, dtype=torch.int32).view(19, 11) * (rank + 1)  # histogram
    w_g = dist.all_reduce(g, async_op=True)
    HF._stat_all_reduce(s)
    w_h = dist.all_reduce(h, group=ag, async_op=True)
    s2 = s.clone()
    HF._stat_all_reduce(s2)
    w_g.wait()
    w_h.wait()
    tri = world * (world + 1) // 2
    ok = ok and bool((g == tri).all())
    ok = ok and bool(torch.equal(s[:, 0], torch.full((7,), float(tri), dtype=torch.float64)))
    ok = ok and bool(torch.equal(s2, s * world))
    ok = ok and bool(torch.equal(h, torch.arange(19 * 11, dtype=torch.int32).view(19, 11) * tri))
    # counted collectives (round 5): comm.all_reduce reduces on the named communicator and counts per communicator — the
    # per-step invariant bench.py prints at N > 1 (6 buckets / 208 statistics reduces / 3 auxiliary reduces) is built on this
    before = dict(comm.COUNTS)
    a = torch.full((4,), float(rank + 1), dtype=torch.float64)
    b = torch.full((4,), float(rank + 1), dtype=torch.float64)
    hnd = comm.all_reduce(a, "stat", async_op=True)
    comm.all_reduce(b, "aux")
    comm.all_reduce(b, "aux")
    hnd.wait()
    ok = ok and bool((a == tri).all()) and bool((b == tri * world).all())
    ok = ok and comm.COUNTS["stat"] - before["stat"] == 1 and comm.COUNTS["aux"] - before["aux"] == 2
    # HIAST_COMM_GROUPS=0: everything on the default group (group=None)
    os.environ["HIAST_COMM_GROUPS"] = "0"
    ok = ok and comm.stat_group() is None and comm.aux_group() is None
    t = torch.ones(3, dtype=torch.float64)
    HF._stat_all_reduce(t)
    ok = ok and bool((t == world).all())
    os.environ["HIAST_COMM_GROUPS"] = "1"
    if rank == 0:
        np.save(out, np.array([ok]))
    dist.barrier()
    dist.destroy_process_group()
    comm.reset()


def test_stat_and_aux_groups_are_separate_communicators(tmp_path):
    """utils/comm.py: the SyncBN sums and the pseudo-label exchange run on process groups of their own (not DDP's
    default communicator); same sums whatever the interleaving; HIAST_COMM_GROUPS=0 falls back to the default group"""
    out = str(tmp_path / "ok.npy")
    mp.spawn(_comm_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert bool(np.load(out)[0])


def _rehearsal_worker(rank, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from hiast_amd.utils import comm
    from hiast_amd import functional as HF
    dist.init_process_group("gloo", rank=0, world_size=1)
    sbn, bn = torch.nn.SyncBatchNorm(4), torch.nn.BatchNorm2d(4)
    os.environ.pop("HIAST_DIST_REHEARSAL", None)
    res = [comm.multi(), comm.stat_group() is None, HF._sync_world(sbn), HF._sync_world(bn)]
    os.environ["HIAST_DIST_REHEARSAL"] = "1"
    res += [comm.multi(), comm.stat_group() is not None and comm.stat_group() is not comm.aux_group(),
            HF._sync_world(sbn), HF._sync_world(bn)]
    before = comm.COUNTS["stat"]
    t = torch.arange(6, dtype=torch.float64)
    HF._stat_wait(HF._stat_all_reduce(t, async_op=True))
    res += [bool(torch.equal(t, torch.arange(6, dtype=torch.float64))), comm.COUNTS["stat"] - before]
    np.save(out, np.array([float(v) for v in res]))
    dist.destroy_process_group()
    comm.reset()


def test_one_rank_rehearsal_takes_the_exchange_path(tmp_path):
    """HIAST_DIST_REHEARSAL=1 (utils/comm.py, `bench.py --rehearse-dist`): with a ONE-rank process group the package takes the
    N > 1 path — own communicators, SyncBN layers report world 0 (= exchange, count of this rank), plain BN stays 1 — and without
    the switch a one-rank group is a single process"""
    out = str(tmp_path / "r.npy")
    mp.spawn(_rehearsal_worker, args=(_free_port(), out), nprocs=1, join=True)
    assert np.load(out).tolist() == [0, 1, 1, 1, 1, 1, 0, 1, 1, 1]


def test_usable_cpus_respects_affinity():
    from hiast_amd.utils import comm
    n = comm.usable_cpus()
    assert 1 <= n <= len(os.sched_getaffinity(0))
