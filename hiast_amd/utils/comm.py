"""Communicators of the N>1 path (one process per GPU, torch.distributed; backend 'nccl' = RCCL over xGMI).

The reference has ONE exchange path (apex DDP + apex SyncBN on the default group, code/train.py:52-59,82,
workflows/trainer/base_trainer.py:43-56, utils/utils.py:103-105).  Here three kinds of traffic share a step:

  * DDP's gradient buckets (175 MB per step in 32 MB all-reduces, overlapped with backward)   -> default group
  * the SyncBN sums: 208 all-reduces of [C,2] doubles per step, each between two kernels of the
    main stream, i.e. latency-critical                                                          -> stat_group()
  * the pseudo-label exchange (19 x 15361 u32 histogram, [C] class sums), validation I/U areas  -> aux_group()

A communicator executes its operations in issue order: on the default group a [C,2] reduce issued during backward
queues behind whatever 32 MB bucket DDP's reducer has issued before it, and the histogram all-reduce (which waits for
the whole pseudo-label forward) holds back every SyncBN reduce issued after it.  Separate process groups have separate
RCCL communicators and streams, so the three kinds only meet on the links.

Every rank must create the groups at the same point of its program: `setup()` is called right after
init_process_group by the trainers / generator / bench.py; the getters create lazily otherwise (first use is at the
same program point on every rank as well).  HIAST_COMM_GROUPS=0 routes everything through the default group.
"""
import datetime
import os
import time

import torch.distributed as dist

_groups = {}


def rehearsal():
    """HIAST_DIST_REHEARSAL=1: a process group of ONE rank takes the N > 1 code path — SyncBN layers exchange their [C,2] sums
    through the statistics communicator, the pseudo-label histogram goes through the auxiliary one, weight gradients stay on the
    main stream under DDP.  A one-rank all-reduce is the identity, so results must match the single-process step; what the
    rehearsal exercises is the BACKEND: `bench.py --rehearse-dist` is the only way to run torch's RCCL process group (its own
    streams, event hand-offs, async work handles) on a one-GPU box (RCCL refuses two ranks on one device)."""
    return os.environ.get("HIAST_DIST_REHEARSAL", "0") == "1"


def multi():
    """does this process run the N > 1 code path?  (more than one rank, or the one-rank rehearsal)"""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or rehearsal())


def _enabled():
    return os.environ.get("HIAST_COMM_GROUPS", "1") != "0" and multi()


def _get(name):
    if not _enabled():
        return None                          # dist.all_reduce(..., group=None) = the default group
    ent = _groups.get(name)
    if ent is None or ent[0] is not dist.distributed_c10d._get_default_group():
        # (a re-initialised default group — tests start several process groups in one interpreter — invalidates ours)
        ent = _groups[name] = (dist.distributed_c10d._get_default_group(), dist.new_group(timeout=timeout()))
    return ent[1]


def setup():
    """create both groups now (collective: every rank, same order).  No-op for a single process."""
    stat_group()
    aux_group()


def stat_group():
    """group of the SyncBN statistics all-reduces (hiast_amd/functional.py)"""
    return _get("stat")


def aux_group():
    """group of the pseudo-label histogram / class-sum exchange and of the validation areas"""
    return _get("aux")


def reset():
    """forget the groups (after destroy_process_group)"""
    _groups.clear()


# Collectives issued per communicator (host-side counters, a few ns each): the invariant a first run on a new backend is
# checked against — per self-training step at N > 1: 2 x (number of SyncBN layers) [C,2] reduces on the statistics group
# (ResNet-101: 208), 3 on the auxiliary group (histogram + the two class sums), DDP's gradient buckets on the default group
# (43.8 M fp32 parameters in 32 MB buckets: 6; counted by the reducer itself, `_get_ddp_logging_data()`).
COUNTS = {"stat": 0, "aux": 0}
# ... and the HOST time spent inside those calls (issue + whatever the backend blocks for; `wait` adds the handle waits of the
# async form): what a first SCALE record is read against — 208 x 20-40 us budgeted for the statistics communicator
HOST_S = {"stat": 0.0, "aux": 0.0}


def all_reduce(t, which, async_op=False, **kw):
    """dist.all_reduce on the `which` ('stat' | 'aux') communicator, counted and host-timed"""
    COUNTS[which] += 1
    t0 = time.perf_counter()
    h = dist.all_reduce(t, group=stat_group() if which == "stat" else aux_group(), async_op=async_op, **kw)
    HOST_S[which] += time.perf_counter() - t0
    return h


def wait(handle, which):
    """handle.wait() of an async all_reduce of the `which` communicator, host-timed"""
    t0 = time.perf_counter()
    handle.wait()
    HOST_S[which] += time.perf_counter() - t0


def timeout():
    """timeout of every process group this package creates (HIAST_DIST_TIMEOUT_S, default 180 s): a collective that one rank
    never joins ends in an exception on the others instead of the backend's 10 / 30 minute default — the first RCCL hang would
    otherwise burn a whole GPU lease and return nothing (reference: dist.init_process_group in code/train.py:52-59 runs on
    torch's default)"""
    return datetime.timedelta(seconds=float(os.environ.get("HIAST_DIST_TIMEOUT_S", "180")))


def init_process_group(backend, **kw):
    """dist.init_process_group(backend, timeout=timeout(), ...) + setup() of the statistics / auxiliary communicators"""
    kw.setdefault("timeout", timeout())
    dist.init_process_group(backend=backend, **kw)
    setup()


def apply_cu_reserve(world=None, device=None):
    """HIAST_RESERVE_CUS=n (default 0 = off), honoured when the process group has more than one rank
    (HIAST_RESERVE_CUS_FORCE=1: also in a single process — measuring what the reserve costs): the library sizes every persistent
    / one-block-per-CU launch to CUs - n (hiast_set_reserve_cus) and the CURRENT stream of this thread becomes a stream whose
    kernels cannot be placed on those n CUs (hiast_stream_create_reserved; side streams: functional.new_stream) — a collective's
    kernel then always finds a CU no 60-160 us tile-kernel block holds (design risk at N > 1).  Call it once, after
    torch.cuda.set_device and before the first launch.  -> the reserve in effect"""
    import torch
    n = int(os.environ.get("HIAST_RESERVE_CUS", "0") or 0)
    if world is None:
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if n <= 0 or (world <= 1 and os.environ.get("HIAST_RESERVE_CUS_FORCE", "0") != "1"):
        return 0
    from hiast_amd import kernels as K
    K.reserve_cus(n)
    n = K.reserve_cus()
    torch.cuda.set_stream(K.reserved_stream(n, device))
    return n


def usable_cpus():
    """CPUs this process may actually use: affinity mask capped by the cgroup quota (an MI355X box shows 256 logical
    CPUs and grants 16; thread pools sized by os.cpu_count() run 10x slower there than pools sized by the quota)"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], txt[1]
            else:
                quota, period = txt[0], open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            if quota not in ("max", "-1") and int(period) > 0:
                n = max(1, min(n, int(int(quota) / int(period))))
                break
        except (OSError, ValueError, IndexError):
            continue
    return n
