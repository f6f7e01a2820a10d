"""PSEUDO_POLICY registry: 'IAS' (instance-adaptive selector), 'CT', 'NT', 'CBST'
(reference: workflows/pseudo_label_generator.py:14-213).

What changed under the same interface (`PSEUDO_POLICY[type](cfg).run()`, same artefacts on disk):
  * the model hands over LOW-RES logits; upsample + softmax + max/argmax + the per-class fp16
    confidence multiset (as a 19 x 15361 integer histogram) is one HIP kernel (pass 1); nothing of
    size B x C x H x W is built and nothing but the histogram (1.2 MB) and the final uint8 label maps
    crosses PCIe;
  * thresholds come from the histogram on the host (hiast_amd/workflows/ias_math.py) — bit-identical
    to the reference's list + np.quantile formulation;
  * select / count / Σprob is a second HIP kernel (pass 2);
  * with torch.distributed initialised, images of a global batch are split over ranks and the
    histograms / class sums are all-reduced (RCCL): the result equals the single-process run with
    batch_size = world * local_batch on the same image order.
"""
import json
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
import torch.distributed as dist
from PIL import Image
from torch.utils.data import DataLoader, Sampler

from hiast_amd import functional as HF
from hiast_amd.utils import comm, utils
from hiast_amd.utils.registry.registries import DATASET, PSEUDO_POLICY
from hiast_amd.workflows import ias_math


class ShardedBatchSampler(Sampler):
    """Global batches of `world * batch_size` consecutive positions of a seeded permutation; rank r
    takes the r-th `batch_size` slice of each (possibly empty at the tail)."""

    def __init__(self, n, batch_size, rank, world, shuffle=True, seed=888):
        self.n, self.bs, self.rank, self.world = n, batch_size, rank, world
        if shuffle:
            g = torch.Generator()
            g.manual_seed(seed)
            self.order = torch.randperm(n, generator=g).tolist()
        else:
            self.order = list(range(n))

    def __len__(self):
        gb = self.bs * self.world
        return (self.n + gb - 1) // gb

    def __iter__(self):
        gb = self.bs * self.world
        for s in range(0, self.n, gb):
            yield self.order[s + self.rank * self.bs: min(s + (self.rank + 1) * self.bs, s + gb, self.n)]


class HipPlabelEngine:
    """Device side of one generator step (the HIP kernels); no CPU path."""

    def __init__(self, model, device, num_classes):
        if torch.device(device).type != "cuda":
            raise RuntimeError("HipPlabelEngine runs on the HIP device only (no CPU path); got %r" % (device,))
        self.model, self.device, self.C = model, device, num_classes

    @staticmethod
    def _equiv(n_images, hw):
        """images of the bench size (1024x512) that n_images of size hw = (h, w) amount to: what fills the chip is pixels"""
        return n_images if not hw else n_images * (float(hw[0]) * float(hw[1])) / (512.0 * 1024.0)

    def lanes(self, batch_size, hw=None):
        """how many batches the pipelined generators keep in flight on forward streams of their own.  A forward over fewer than
        8 images leaves most of the chip idle (a layer3 launch at batch 2 has 64 tiles of 256 rows, 128 as half tiles, for 256
        CUs: 2.85 ms per image against 1.92 at batch 8, profiles/r06_generator_bs2_kernel_stats.csv) and pass 1 of a batch does
        not depend on the thresholds of the batch before it — only pass 2 does — so the forwards of batch t + 1 and t + 2 run
        SIDE BY SIDE on two streams (round 6; HIAST_GEN_LANES=1: one after the other as before).  Batches of 8 or more images
        already run as two sub-batches on two streams (functional.eval_forward_split)."""
        n = int(os.environ.get("HIAST_GEN_LANES", "2"))
        return max(1, n) if (batch_size < 8 and self._equiv(batch_size, hw) < 8) else 1

    def group(self, batch_size, hw=None):
        """how many CONSECUTIVE loader batches share one forward (round 6).  An inference forward treats every image alone
        (BatchNorm in eval mode) and pass 1 of a batch does not depend on the thresholds of the batch before it, so the trunk runs
        ONCE over the images of several small batches (a layer3 launch over two images fills a quarter of the chip: 2.85 ms per
        image against 1.92 at batch 8) and pass 1 runs per original batch on its slice of the logits: every batch keeps its own
        histogram, the threshold recursion advances batch by batch exactly as before, label maps and statistics are bit for bit
        those of the one-batch-at-a-time loop — the tile kernels' outputs do not depend on the launch size (same products in the
        same order per element), PROVIDED the launch size does not change which kernel runs: the register-resident-weight kernels
        (xconv / xconv2) take a 1x1 launch from 4096 output rows on and differ from the tile kernel in summation order (2e-5).
        Frames whose 1/8-resolution map has fewer than 4096 pixels (below 512 x 512) would cross that gate by being grouped, so
        they are not grouped (nor is anything when the frame size is unknown).  HIAST_GEN_GROUP=1: one forward per loader batch."""
        if not hw or (int(hw[0]) // 8) * (int(hw[1]) // 8) < 4096:
            return 1
        env = os.environ.get("HIAST_GEN_GROUP", "")
        # measured (profiles/r06_generator_lanes.txt): forwards over 4 images, two of them side by side (lanes), beat forwards
        # over 6 or 8 images at every small batch size — the smallest group that reaches 4 images
        # (images of 1024x512; larger frames count by their pixels: a 2048x1024 frame is four of them and needs no company)
        import math
        g = int(env) if env else int(math.ceil(float(os.environ.get("HIAST_GEN_GROUP_IMAGES", "4")) /
                                               max(1e-9, self._equiv(int(batch_size), hw)) - 1e-9))
        return max(1, min(g, 8))

    @torch.no_grad()
    def begin_group(self, batches, lane=None):
        """begin() for several consecutive loader batches with ONE forward over their images -> one state per batch (in order)"""
        if len(batches) == 1:
            return [self.begin(batches[0], lane)]
        live = [b for b in batches if b is not None and b.shape[0] > 0]
        if len(live) <= 1 or len({tuple(b.shape[1:]) for b in live}) != 1 or len({b.dtype for b in live}) != 1:
            return [self.begin(b, lane) for b in batches]          # (ragged image sizes: no shared launch)
        if lane is None:
            return self._begin_group(batches, live, False)
        self._lane_ready(lane)
        with torch.cuda.stream(self._lane_stream(lane)):
            return self._begin_group(batches, live, True)

    def _begin_group(self, batches, live, side_by_side):
        import contextlib
        from hiast_amd import kernels as K
        imgs = torch.cat([b.to(self.device, non_blocking=True) for b in live], 0)
        if imgs.dtype == torch.uint8:
            from hiast_amd.sseg.datasets.utils import MEAN, STD
            imgs = K.normalize_u8(imgs, MEAN, STD)
        if getattr(self, "_fwd", None) is None or self._fwd.model is not self.model:
            self._fwd = HF.GraphedEval(self.model, None)
        H, W = imgs.shape[2:]
        with (K.cosched() if side_by_side else contextlib.nullcontext()):
            logits = self._fwd(imgs).contiguous()
        states, i0 = [], 0
        for b in batches:
            st = {"mp": None, "am": None}
            if b is None or b.shape[0] == 0:
                st["hist"] = torch.zeros((self.C, ias_math.NBINS), dtype=torch.int32, device=self.device)
            else:
                n = b.shape[0]
                st["mp"], st["am"], st["hist"] = K.plabel_pass1(logits[i0:i0 + n], H, W)     # this batch's own histogram
                i0 += n
            states.append(st)
        ev = torch.cuda.Event()
        ev.record()
        for st in states:
            st["ev"] = ev
        return states

    def _lane_ready(self, lane):
        # the kernel-format weight copies are packed ONCE, on the calling stream, before any lane reads them (the model does not
        # change during generation); a lane waits for that event at its first use only — waiting for the calling stream every time
        # would put the lanes back in single file
        if getattr(self, "_packed_ev", None) is None:
            HF.prepack_eval_trunks(self.model, torch.empty((1, 3, 8, 8), dtype=torch.float32, device=self.device))
            self._packed_ev = torch.cuda.Event()
            self._packed_ev.record()
            self._lane_seen = set()
        if lane not in self._lane_seen:
            self._lane_stream(lane).wait_event(self._packed_ev)
            self._lane_seen.add(lane)

    def _lane_stream(self, lane):
        ls = self.__dict__.setdefault("_lanes", {})
        if lane not in ls:
            ls[lane] = HF.new_stream(self.device)
        return ls[lane]

    @torch.no_grad()
    def begin(self, imgs, lane=None):
        """forward + pass 1 of one batch, enqueued on the current stream (lane None) or on forward stream `lane`; nothing waits.
        -> state for hist_host() / finish()."""
        if lane is None or imgs is None or imgs.shape[0] == 0:
            return self._begin(imgs, False)
        self._lane_ready(lane)
        with torch.cuda.stream(self._lane_stream(lane)):
            return self._begin(imgs, True)

    def _begin(self, imgs, side_by_side):
        """forward + pass 1 of one batch on the current stream.
        The generators keep the device busy with the NEXT batch's begin() while the host turns this batch's histogram into
        thresholds and its label maps into PNG files; everything that follows pass 1 of a batch (histogram exchange and
        read-back, pass 2, read-back of the maps) runs on a second stream behind the state's event, so it does not queue up
        behind the next forward."""
        from hiast_amd import kernels as K
        C = self.C
        st = {"mp": None, "am": None}
        if imgs is None or imgs.shape[0] == 0:
            st["hist"] = torch.zeros((C, ias_math.NBINS), dtype=torch.int32, device=self.device)
        else:
            imgs = imgs.to(self.device, non_blocking=True)
            if imgs.dtype == torch.uint8:        # dataset.device_transform: ToTensor + Normalize here, on the device
                from hiast_amd.sseg.datasets.utils import MEAN, STD
                imgs = K.normalize_u8(imgs, MEAN, STD)
            from hiast_amd import functional as HF
            if getattr(self, "_fwd", None) is None or self._fwd.model is not self.model:
                # the fp32-class inference forward (two sub-batches on two streams for 8 or more images); HIAST_GRAPH_EVAL=1:
                # replayed from a captured HIP graph once a batch shape has been seen twice (no gain once the batches are
                # pipelined: 473 vs 487 images/s)
                self._fwd = HF.GraphedEval(self.model, None)
            H, W = imgs.shape[2:]
            import contextlib
            # (two forwards side by side: half-chip launches keep their 256-row tile form, as for the sub-batches of
            # eval_forward_split — the thread-local hint of the library)
            with (K.cosched() if side_by_side else contextlib.nullcontext()):
                logits = self._fwd(imgs).contiguous()
            st["mp"], st["am"], st["hist"] = K.plabel_pass1(logits, H, W)
        st["ev"] = torch.cuda.Event()
        st["ev"].record()
        return st

    def post_stream(self):
        """the stream of everything behind pass 1 (context manager target)"""
        if getattr(self, "_post", None) is None:
            self._post = HF.new_stream(self.device)
        return self._post

    @torch.no_grad()
    def hist_host(self, st, allreduce=None):
        """the (all-reduced) histogram of a begin() state on the host: uint32 view [C, NBINS]"""
        post = self.post_stream()
        with torch.cuda.stream(post):
            post.wait_event(st["ev"])
            h = st["hist"] if allreduce is None else allreduce(st["hist"])
            return h.cpu().numpy().view(np.uint32)      # the copy runs on (and synchronises) the post stream

    @torch.no_grad()
    def finish(self, st, thr64):
        """pass 2 of a begin() state on the post stream -> (plbl uint8 device tensor or None, count, sumprob_fx); the
        caller reads them back inside `with torch.cuda.stream(engine.post_stream())`"""
        from hiast_amd import kernels as K
        post = self.post_stream()
        with torch.cuda.stream(post):
            post.wait_event(st["ev"])
            if st["mp"] is None:
                z = torch.zeros((self.C,), dtype=torch.int64, device=self.device)
                return None, torch.zeros((0, self.C), dtype=torch.int64, device=self.device), z
            thr_up = None if thr64 is None else K.h2d_async(ias_math.roundup_f32(thr64), self.device)
            return K.plabel_pass2(st["mp"], st["am"], thr_up, self.C)

    @torch.no_grad()
    def pass1(self, imgs):
        """one batch at a time (the constant-threshold policies): begin() with the state kept for pass2() / strided_hist()"""
        st = self.begin(imgs)
        self._mp, self._am = st["mp"], st["am"]
        return st["hist"]

    @torch.no_grad()
    def strided_hist(self, interval, rank_offset=None):
        """CBST's sample of the batch pass1() has just seen: every `interval`-th pixel of each class in raster order
        (offset by the class pixels lower ranks hold in the same global batch) -> hist i32 [C,NBINS]"""
        from hiast_amd import kernels as K
        if self._mp is None:
            return torch.zeros((self.C, ias_math.NBINS), dtype=torch.int32, device=self.device)
        off = None if rank_offset is None else torch.as_tensor(rank_offset, dtype=torch.int64).to(self.device)
        return K.plabel_strided_hist(self._mp, self._am, self.C, interval, rank_offset=off)

    @torch.no_grad()
    def pass2(self, thr64):
        """-> (plbl uint8 numpy [b,H,W] or None, count i64 tensor [b,C], sumprob_fx i64 tensor [C])"""
        from hiast_amd import kernels as K
        C = self.C
        if self._mp is None:
            z = torch.zeros((C,), dtype=torch.int64, device=self.device)
            return None, torch.zeros((0, C), dtype=torch.int64, device=self.device), z
        thr_up = None if thr64 is None else K.h2d_async(ias_math.roundup_f32(thr64), self.device)
        plbl, count, sfx = K.plabel_pass2(self._mp, self._am, thr_up, C)
        return plbl, count, sfx


class BasePseudoGenerator:

    def __init__(self, cfg, engine=None, dataset=None):
        self.cfg = cfg
        if torch.cuda.is_available():
            utils.limit_cpu_threads()
        C = cfg.dataset.num_classes
        self.statics_class = np.zeros(C, dtype=np.int64)
        self.sample_stats = []
        self.samples_class = {i: [] for i in range(C)}
        self.class_mean_probs = np.zeros(C)
        self.class_threshold = None
        self.rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.multi = self.world > 1 or comm.multi()      # (the one-rank rehearsal takes the exchange path too: utils/comm.py)
        self._io = ThreadPoolExecutor(max_workers=max(2, cfg.dataset.num_workers))
        self._pending = []
        comm.setup()
        self.initialize(engine, dataset)

    # -- setup -------------------------------------------------------------------------------
    def initialize(self, engine=None, dataset=None):
        pp = self.cfg.pseudo_policy
        if engine is None and not torch.cuda.is_available():
            raise RuntimeError("pseudo-label generation runs on the HIP device; no GPU is visible "
                               "and there is no CPU fallback")
        # The DataLoader (and its worker processes) come FIRST, the model goes to the device afterwards: workers forked
        # from a process that already holds device memory and queues stall its first device work by ~1.8 s on MI355X
        # (measured, tools/dbg/generator_batch_times.py: every copy-on-write fault of a starting worker on memory the
        # driver has registered evicts and restores the parent's queues) — and this way the workers decode the first
        # batches while the checkpoint is read and copied to the device.
        if dataset is None:
            aug_type = ["PRS-{}-{}".format(pp.resize_size[0], pp.resize_size[1])]
            tgt = self.cfg.dataset.target
            dataset = DATASET[tgt.type](self.cfg, tgt.json_path, tgt.image_dir, aug_type=aug_type,
                                        num_classes=self.cfg.dataset.num_classes)
        self.t_dataset = dataset
        # workers hand over uint8 images; normalisation runs on the device (same bits, 4x less host traffic)
        dataset.device_transform = ((engine is None or isinstance(engine, HipPlabelEngine))
                                    and os.environ.get("HIAST_HOST_TRANSFORM", "0") != "1")
        sampler = ShardedBatchSampler(len(dataset), pp.batch_size, self.rank, self.world, shuffle=True,
                                      seed=self.cfg.train.random_seed)
        nw = self.cfg.dataset.num_workers
        self.t_loader = DataLoader(dataset, batch_sampler=sampler, num_workers=nw, pin_memory=torch.cuda.is_available(),
                                   collate_fn=_collate, persistent_workers=nw > 0)      # (CBST walks the set twice)
        self._first_iter = iter(self.t_loader) if (engine is None and nw > 0) else None
        if engine is None:
            device = utils.get_device()
            model = utils.load_model(self.cfg, resume_from=pp.resume_from).to(device).eval()
            engine = HipPlabelEngine(model, device, self.cfg.dataset.num_classes)
        self.engine = engine
        self.pseudo_label_save_dir = pp.save_dir
        assert self.pseudo_label_save_dir is not None and (
            not os.path.exists(self.pseudo_label_save_dir) or len(os.listdir(self.pseudo_label_save_dir)) == 0
            or self.rank != 0)
        if self.rank == 0:
            os.makedirs(self.pseudo_label_save_dir, exist_ok=True)
        if self.multi:
            dist.barrier()

    # -- artefacts ---------------------------------------------------------------------------
    def save_pseudo_label(self, plbl, img_path):
        name = os.path.splitext(os.path.basename(img_path))[0]
        path = os.path.join(self.pseudo_label_save_dir, "{}_pseudo_label.png".format(name))
        self._pending.append(self._io.submit(lambda a=np.ascontiguousarray(plbl, dtype=np.uint8), p=path:
                                             Image.fromarray(a, mode="L").save(p, compress_level=1)))

    def save_data(self):
        for f in self._pending:
            f.result()
        self._pending = []
        if self.multi:   # per-image records live on their owner rank: gather them on rank 0
            parts = [None] * self.world
            dist.all_gather_object(parts, (self.sample_stats, self.samples_class))
            self.sample_stats = [s for p in parts for s in p[0]]
            merged = {i: [] for i in range(self.cfg.dataset.num_classes)}
            for p in parts:
                for k, v in p[1].items():
                    merged[int(k)].extend(v)
            self.samples_class = merged
        if self.rank != 0:
            return
        root = os.path.join(self.pseudo_label_save_dir, "..")
        if self.class_threshold is not None:
            print("class threshold: {}".format(self.class_threshold))
            np.save(os.path.join(root, "class_threshold.npy"), self.class_threshold)
        print("class statics number: {}".format(self.statics_class))
        np.save(os.path.join(root, "statics_class.npy"), self.statics_class)
        print("class mean probabilities: {}".format(self.class_mean_probs))
        np.save(os.path.join(root, "class_mean_probabilities.npy"), self.class_mean_probs)
        with open(os.path.join(root, "sample_class_stats.json"), "a") as f:   # append mode, as the reference
            f.write(json.dumps(self.sample_stats))
        with open(os.path.join(root, "samples_with_class.json"), "a") as f:
            f.write(json.dumps(self.samples_class))

    # -- one batch ---------------------------------------------------------------------------
    def _allreduce(self, t):
        if self.multi:      # histogram / class sums: on the auxiliary communicator (utils/comm.py)
            comm.all_reduce(t, "aux", op=dist.ReduceOp.SUM)
        return t

    def select_and_save_confident_label(self, img_paths, state=None):
        """pass 2 + statistics of select_and_save_confident_label (pseudo_label_generator.py:67-105).
        state: a HipPlabelEngine.begin() state (pipelined generators) instead of the engine's last pass1() batch"""
        import contextlib
        if state is None:
            plbl, count, sfx = self.engine.pass2(self.class_threshold)
            ctx = contextlib.nullcontext()
        else:
            plbl, count, sfx = self.engine.finish(state, self.class_threshold)
            ctx = torch.cuda.stream(self.engine.post_stream())     # exchanges and read-backs on the stream that produced them
        with ctx:
            count_c = self._allreduce(count.sum(0) if count.shape[0] else torch.zeros_like(sfx))
            sfx = self._allreduce(sfx)
            count_h = count.cpu().numpy()
            plbl_h = plbl.cpu().numpy() if plbl is not None else None
            count_c = count_c.cpu().numpy()
            sfx_h = sfx.cpu().numpy()
        C = self.cfg.dataset.num_classes
        for b, path in enumerate(img_paths):
            stats = {}
            for i in range(C):
                n = int(count_h[b, i])
                if n != 0:
                    stats[i] = n
                    self.samples_class[i].append([path, n])
            stats["file"] = path
            self.sample_stats.append(stats)
            self.save_pseudo_label(plbl_h[b], path)
        self.statics_class += count_c
        ias_math.update_class_mean_probs(self.class_mean_probs, count_c, sfx_h, self.cfg.preprocessor.copy_paste.gamma)
        return plbl_h

    def _batches(self):
        first, self._first_iter = self._first_iter, None        # the iterator initialize() has started, once
        for data in (first if first is not None else self.t_loader):
            if data is None:
                yield None, []
            else:
                yield data["images"], list(data["image_paths"])

    def _pipelined_states(self):
        """(image paths, HipPlabelEngine.begin() state) per batch, with forward + pass 1 of the NEXT batch(es) already enqueued
        when a batch is handed out: the device works on batch t+1 (small batches: t+1 and t+2 side by side on two forward
        streams, HipPlabelEngine.lanes) while the host finishes batch t.  The order in which batches are handed out — and with it
        the threshold recursion and every artefact — is the loader's, whatever the depth."""
        from collections import deque
        batches = iter(self._batches())
        bs = int(self.cfg.pseudo_policy.batch_size)
        hw = tuple(self.cfg.pseudo_policy.resize_size) if self.cfg.pseudo_policy.resize_size else None
        grp = self.engine.group(bs, hw) if hasattr(self.engine, "group") else 1          # loader batches per forward
        depth = self.engine.lanes(bs * grp, hw) if hasattr(self.engine, "lanes") else 1  # forwards in flight on streams of their own
        q = deque()
        n = 0

        def enqueue():
            nonlocal n
            items = []
            while len(items) < grp:
                nxt = next(batches, None)
                if nxt is None:
                    break
                items.append(nxt)
            if not items:
                return False
            lane = n % depth if depth > 1 else None
            if grp > 1:
                sts = self.engine.begin_group([it[0] for it in items], lane)
            else:
                sts = [self.engine.begin(items[0][0], lane) if depth > 1 else self.engine.begin(items[0][0])]
            q.append([(it[1], st) for it, st in zip(items, sts)])
            n += 1
            return True
        for _ in range(depth):
            if not enqueue():
                break
        while q:
            group = q.popleft()
            enqueue()
            for paths, st in group:
                yield paths, st

    def _exists(self):
        """rank 0 looks at the directory; every rank takes ITS decision (a rank that went on alone into the
        collectives of run() would wait for the others forever)"""
        done = self.rank == 0 and len(os.listdir(self.pseudo_label_save_dir)) >= len(self.t_dataset)
        if self.multi:
            box = [bool(done)]
            dist.broadcast_object_list(box, src=0)
            done = box[0]
        if done and self.rank == 0:
            print("%% pseudo labels have existed")
        return bool(done)

    def run(self):
        raise NotImplementedError


def _collate(items):
    if len(items) == 0:
        return None
    return {"images": torch.stack([it["images"] for it in items]),
            "image_paths": [it["image_paths"] for it in items]}


@PSEUDO_POLICY.register("CT")
class ConstantThresholdPseudoGenerator(BasePseudoGenerator):

    def get_constant_threshold(self):
        return self.cfg.pseudo_policy.ct.threshold * np.ones(self.cfg.dataset.num_classes)

    def run(self):
        if self._exists():
            return
        self.class_threshold = self.get_constant_threshold()
        if hasattr(self.engine, "begin"):       # (test doubles have the two-call interface only)
            for paths, st in self._pipelined_states():
                self.select_and_save_confident_label(paths, st)
        else:
            for imgs, paths in self._batches():
                self.engine.pass1(imgs)
                self.select_and_save_confident_label(paths)
        self.save_data()


@PSEUDO_POLICY.register("NT")
class NoThresholdPseudoGenerator(ConstantThresholdPseudoGenerator):

    def get_constant_threshold(self):
        return None


@PSEUDO_POLICY.register("CBST")
class CBSTPseudoGenerator(ConstantThresholdPseudoGenerator):
    """Global per-class quantile over the whole target set (pseudo_label_generator.py:142-165): per batch and class,
    every `sample_interval`-th confidence of the class's pixels in raster order enters the pool (here: the pooled
    fp16 histogram); the threshold is its (1 - p) quantile.  Sharded runs rank the pixels of a global batch across the
    ranks (rank r's pixels follow those of ranks < r), so the sample equals the single-process one at
    batch_size = world x local.  A class that is never predicted gets NaN (numpy 1.19's np.quantile([]); numpy >= 1.22
    raises there) — no pixel carries that label, so it is never compared."""

    def get_constant_threshold(self):
        C = self.cfg.dataset.num_classes
        interval = int(self.cfg.pseudo_policy.cbst.sample_interval)
        total = torch.zeros((C, ias_math.NBINS), dtype=torch.int64, device=self.engine.device)
        for imgs, _ in self._batches():
            full = self.engine.pass1(imgs)
            offset = None
            if self.multi:          # class-c pixels of this global batch held by the lower ranks
                mine = full.long().sum(1)
                parts = [torch.zeros_like(mine) for _ in range(self.world)]
                dist.all_gather(parts, mine, group=comm.aux_group())
                offset = torch.stack(parts[:self.rank]).sum(0) if self.rank else torch.zeros_like(mine)
            total += (full if interval == 1 and offset is None else self.engine.strided_hist(interval, offset)).long()
        total = self._allreduce(total)
        return ias_math.cbst_threshold(total.cpu().numpy(), self.cfg.pseudo_policy.cbst.p)


@PSEUDO_POLICY.register("IAS")
class IASPseudoGenerator(BasePseudoGenerator):

    def run(self):
        if self._exists():
            return
        ias = self.cfg.pseudo_policy.ias
        self.class_threshold = 0.9 * np.ones(self.cfg.dataset.num_classes)     # :185
        if not hasattr(self.engine, "begin"):           # engines with the two-call interface only (test doubles)
            for imgs, paths in self._batches():
                hist = self._allreduce(self.engine.pass1(imgs))
                _, self.class_threshold = ias_math.ias_update(hist.cpu().numpy().view(np.uint32), self.class_threshold,
                                                              ias.alpha, ias.beta, ias.gamma)
                self.select_and_save_confident_label(paths)
            self.save_data()
            return
        # Software pipeline over the batches: forward + pass 1 of batch t+1 are enqueued BEFORE the host waits for the
        # histogram of batch t.  The threshold recursion (thr_t from thr_{t-1} and hist_t, :185-205) is untouched — pass 1
        # does not depend on the thresholds, only pass 2 does — so every artefact is what the one-batch-at-a-time loop
        # writes; the device no longer idles while the host computes thresholds, reads the label maps back and hands them
        # to the PNG writers, and the host no longer waits through a forward it could have enqueued earlier.
        for paths, st in self._pipelined_states():
            hist = self.engine.hist_host(st, self._allreduce if self.multi else None)
            _, self.class_threshold = ias_math.ias_update(hist, self.class_threshold, ias.alpha, ias.beta, ias.gamma)
            self.select_and_save_confident_label(paths, st)
        self.save_data()
