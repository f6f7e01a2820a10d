"""Tensor-level entry points of the HIP library: validate shapes/dtypes/devices on the host
(a kernel that faults can take the whole node down), allocate outputs with torch, and enqueue
on torch's current HIP stream.  No torch arithmetic happens here and there is no CPU path:
every function requires CUDA(HIP) tensors and raises otherwise.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from . import switches as _SW
from ._lib import NBINS, check


def _stream():
    # raw handle of torch's current stream on the current device; torch.cuda.current_stream().cuda_stream builds two
    # Python objects per call (8.6 us measured, 436 calls per training step)
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def h2d_async(host, device, out=None):
    """Upload a small numpy array without stalling the launch queue.  A copy from PAGEABLE host memory is synchronous
    on ROCm even when asked to be non-blocking — the runtime waits for the stream to reach the copy, i.e. the host
    loses its whole lead over the device (measured in bench.py: 0.7-0.8 ms of idle device after every such copy) —
    so the bytes go through a pinned staging block of torch's caching host allocator, which keeps the block alive
    until the copy has run.  -> device tensor (`out`, filled, if given)"""
    stage = torch.from_numpy(np.ascontiguousarray(host)).pin_memory()
    if out is None:
        return stage.to(device, non_blocking=True)
    out.copy_(stage.view(out.dtype).reshape(out.shape) if stage.dtype != out.dtype else stage.reshape(out.shape),
              non_blocking=True)
    return out


def _req(t, dtype, ndim=None, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.HiastLibraryError("%s must be a CUDA(HIP) tensor: the HIP path has no CPU fallback" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if ndim is not None and t.dim() != ndim:
        raise ValueError("%s must have %d dims, got %s" % (name, ndim, tuple(t.shape)))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t


# ------------------------------------------------------------------------------- launch hints and device geometry
class cosched:
    """`with K.cosched():` — the calling THREAD runs two launch sequences side by side on two streams: half-chip tile-kernel
    launches keep their 256-row form (hiast_igemm_set_cosched; thread-local in the library, never the environment)."""

    def __init__(self, on=True):
        self.on = 1 if on else 0

    def __enter__(self):
        self.prev = _lib.load().hiast_igemm_set_cosched(self.on)
        return self

    def __exit__(self, *exc):
        _lib.load().hiast_igemm_set_cosched(self.prev)
        return False


class force_half_tile:
    """`with K.force_half_tile(v):` (A/B and tests) — v = 1: the 128 x 128 / two-blocks-per-CU tile form wherever it exists,
    0: never, None: automatic (hiast_igemm_set_half; thread-local)."""

    def __init__(self, v):
        self.v = -1 if v is None else int(bool(v))

    def __enter__(self):
        self.prev = _lib.load().hiast_igemm_set_half(self.v)
        return self

    def __exit__(self, *exc):
        _lib.load().hiast_igemm_set_half(self.prev)
        return False


def device_cus():
    """compute units of the current device (hipDeviceGetAttribute through the C ABI; 256 on MI355X)"""
    n = _lib.load().hiast_device_cus()
    if n <= 0:
        raise _lib.HiastLibraryError("hiast_device_cus: no HIP device visible")
    return n


def reserve_cus(n=None):
    """n given: size every one-block-per-CU / persistent launch to (CUs - n) from now on (rounded up to a multiple of 8;
    hiast_set_reserve_cus) -> the previous value; n None: the current reserve."""
    lib = _lib.load()
    if n is None:
        return lib.hiast_get_reserve_cus()
    prev = lib.hiast_set_reserve_cus(int(n))
    if prev < 0:
        check(prev, "hiast_set_reserve_cus")
    return prev


def grid_cus():
    """CUs the persistent launches fill = device_cus() - reserve_cus()"""
    return max(8, device_cus() - reserve_cus())


_reserved_streams = []          # (handle, torch.cuda.ExternalStream): kept alive for the life of the process


def reserved_stream(reserve, device=None):
    """a torch stream on which no kernel can be placed on `reserve` of the CUs (hiast_stream_create_reserved: queue CU mask,
    reserve / 8 CUs of every XCD) — the non-persistent tile kernels cannot size a grid below their tile count"""
    h = ctypes.c_void_p(0)
    check(_lib.load().hiast_stream_create_reserved(ctypes.byref(h), int(reserve)), "hiast_stream_create_reserved")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    st = torch.cuda.ExternalStream(h.value, device=dev)
    _reserved_streams.append((h, st))
    return st


# ------------------------------------------------------------------------------- K2 upsample
def upsample_bilinear_ac_fwd(x, H, W):
    _req(x, torch.float32, 4, "x")
    B, C, h, w = x.shape
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
    if out.numel():
        check(_lib.load().hiast_upsample_bilinear_ac_fwd(_ptr(x), _ptr(out), B, C, h, w, H, W, _stream()),
              "hiast_upsample_bilinear_ac_fwd")
    return out


def upsample_bilinear_ac_bwd(gout, h, w):
    _req(gout, torch.float32, 4, "gout")
    B, C, H, W = gout.shape
    gin = torch.empty((B, C, h, w), dtype=torch.float32, device=gout.device)
    if gin.numel():
        check(_lib.load().hiast_upsample_bilinear_ac_bwd(_ptr(gout), _ptr(gin), B, C, h, w, H, W, _stream()),
              "hiast_upsample_bilinear_ac_bwd")
    return gin


# ------------------------------------------------------------------------------- K3/K4 pseudo labels
_pass1_ws = {}


def plabel_pass1(logits_lr, H, W, hist=None):
    """-> (maxprob f32 [B,H,W], argmax u8 [B,H,W], hist u32-as-int32 [C,NBINS] (accumulated))"""
    _req(logits_lr, torch.float32, 4, "logits_lr")
    B, C, h, w = logits_lr.shape
    dev = logits_lr.device
    maxprob = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    argmax = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    if hist is None:
        hist = torch.zeros((C, NBINS), dtype=torch.int32, device=dev)
    else:
        _req(hist, torch.int32, 2, "hist")
        assert tuple(hist.shape) == (C, NBINS)
    if B:
        lib = _lib.load()
        # scattered counting copy of the histogram (hot counters on different memory lines): one per device and stream,
        # zeroed by the call; a second launch adds it into hist
        # (keyed by the device INDEX — torch.device("cuda") and ("cuda", 0) are different keys — and the stream handle; the
        # table holds at most a few streams per device, so a recycled handle at worst reuses a buffer of the right size)
        key = (logits_lr.get_device(), torch.cuda.current_stream(dev).cuda_stream, C)
        ws = _pass1_ws.get(key)
        if len(_pass1_ws) > 16 and ws is None:
            _pass1_ws.clear()
        if ws is None:
            ws = torch.empty(lib.hiast_plabel_pass1_workspace_bytes(C) // 4, dtype=torch.int32, device=dev)
            _pass1_ws[key] = ws
        check(lib.hiast_plabel_pass1(_ptr(logits_lr), B, C, h, w, H, W, _ptr(maxprob), _ptr(argmax), _ptr(hist), _ptr(ws),
                                     ws.numel() * 4, _stream()), "hiast_plabel_pass1")
    return maxprob, argmax, hist


def plabel_pass2(maxprob, argmax, thr_up, C, count=None, sumprob_fx=None):
    """thr_up: f32 [C] device tensor or None.  -> (plbl u8, count i64 [B,C], sumprob_fx i64 [C])"""
    _req(maxprob, torch.float32, 3, "maxprob")
    _req(argmax, torch.uint8, 3, "argmax")
    assert maxprob.shape == argmax.shape
    B = maxprob.shape[0]
    HW = maxprob.shape[1] * maxprob.shape[2]
    dev = maxprob.device
    if thr_up is not None:
        _req(thr_up, torch.float32, 1, "thr_up")
        assert thr_up.numel() == C
    plbl = torch.empty_like(argmax)
    if count is None:
        count = torch.zeros((B, C), dtype=torch.int64, device=dev)
    if sumprob_fx is None:
        sumprob_fx = torch.zeros((C,), dtype=torch.int64, device=dev)
    _req(count, torch.int64, 2, "count")
    _req(sumprob_fx, torch.int64, 1, "sumprob_fx")
    assert tuple(count.shape) == (B, C) and sumprob_fx.numel() == C
    if B and HW:
        check(_lib.load().hiast_plabel_pass2(_ptr(maxprob), _ptr(argmax), _ptr(thr_up), B, C, HW, _ptr(plbl),
                                             _ptr(count), _ptr(sumprob_fx), _stream()), "hiast_plabel_pass2")
    return plbl, count, sumprob_fx


_strided_ws = {}


def plabel_strided_hist(maxprob, argmax, C, interval, hist=None, rank_offset=None, want_totals=False):
    """CBST's confidence sample of ONE batch (hiast_plabel_strided_hist): every `interval`-th pixel of each class in
    raster order over the batch -> hist int32 [C,NBINS] (accumulated).  rank_offset i64 [C]: class pixels of the
    same global batch on lower ranks.  -> hist, or (hist, class_total i64 [C])"""
    _req(maxprob, torch.float32, None, "maxprob")
    _req(argmax, torch.uint8, None, "argmax")
    assert maxprob.shape == argmax.shape
    dev = maxprob.device
    N = maxprob.numel()
    if hist is None:
        hist = torch.zeros((C, NBINS), dtype=torch.int32, device=dev)
    _req(hist, torch.int32, 2, "hist")
    assert tuple(hist.shape) == (C, NBINS)
    tot = torch.zeros((C,), dtype=torch.int64, device=dev) if want_totals else None
    if rank_offset is not None:
        _req(rank_offset, torch.int64, 1, "rank_offset")
        assert rank_offset.numel() == C
    if N:
        lib = _lib.load()
        n = lib.hiast_plabel_strided_hist_workspace_bytes(N, C)
        ws = _strided_ws.get(dev)
        if ws is None or ws.numel() * 4 < n:
            ws = torch.empty((n + 3) // 4, dtype=torch.int32, device=dev)
            _strided_ws[dev] = ws
        check(lib.hiast_plabel_strided_hist(_ptr(maxprob), _ptr(argmax), N, int(C), int(interval), _ptr(rank_offset),
                                            _ptr(tot), _ptr(hist), _ptr(ws), ws.numel() * 4, _stream()),
              "hiast_plabel_strided_hist")
    return (hist, tot) if want_totals else hist


TTA_FUSED_CLASSES = (19, 16, 9, 2)       # class counts hiast_tta_fused is instantiated for (plabel2.hip)
TTA_FUSED_MAX_SCALES = 8                 # HIAST_TTA_MAX_SCALES of include/hiast_hip.h


def tta_fused_supported(C, n_scales):
    """whether hiast_tta_fused covers this class count / number of scales (it returns HIAST_E_RANGE otherwise)"""
    return int(C) in TTA_FUSED_CLASSES and 1 <= int(n_scales) <= TTA_FUSED_MAX_SCALES


def tta_fused(zs, zfs, sizes, H, W, want_probs=False, want_label=True):
    """Validator.get_multi_scale_and_flip_logits (+ argmax) from the low-res head outputs of every (scale, flip)
    forward: zs / zfs lists of fp32 [B,C,hs,ws] (zfs None = no flip), sizes list of (Hs, Ws) the image was resized to
    -> (probsum fp32 [B,C,H,W] or None, label u8 [B,H,W] or None)"""
    n = len(zs)
    assert n == len(sizes) and n > 0
    B, C = zs[0].shape[:2]
    for z in zs:
        _req(z, torch.float32, 4, "z")
        assert tuple(z.shape[:2]) == (B, C)
    if zfs is not None:
        assert len(zfs) == n
        for z, zf in zip(zs, zfs):
            _req(zf, torch.float32, 4, "zf")
            assert zf.shape == z.shape
    dev = zs[0].device
    vp = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    iv = lambda vs: (ctypes.c_int * n)(*[int(v) for v in vs])
    probs = torch.empty((B, C, H, W), dtype=torch.float32, device=dev) if want_probs else None
    label = torch.empty((B, H, W), dtype=torch.uint8, device=dev) if want_label else None
    check(_lib.load().hiast_tta_fused(vp(zs), vp(zfs) if zfs is not None else None, iv([z.shape[2] for z in zs]),
                                      iv([z.shape[3] for z in zs]), iv([s[0] for s in sizes]), iv([s[1] for s in sizes]),
                                      n, B, C, int(H), int(W), _ptr(probs), _ptr(label), _stream()), "hiast_tta_fused")
    return probs, label


# ------------------------------------------------------------------------------- K15 discriminator input
def dinput_fwd(logits_lr, H, W, entropy):
    """low-res logits [B,C,h,w] -> softmax (or weighted self-information) map [B,C,H,W] of the upsampled logits"""
    _req(logits_lr, torch.float32, 4, "logits_lr")
    B, C, h, w = logits_lr.shape
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=logits_lr.device)
    if out.numel():
        check(_lib.load().hiast_dinput_fwd(_ptr(logits_lr), int(bool(entropy)), _ptr(out), B, C, h, w, H, W, _stream()),
              "hiast_dinput_fwd")
    return out


def dinput_bwd(logits_lr, gout, entropy):
    _req(logits_lr, torch.float32, 4, "logits_lr")
    _req(gout, torch.float32, 4, "gout")
    B, C, h, w = logits_lr.shape
    assert tuple(gout.shape[:2]) == (B, C)
    H, W = gout.shape[2:]
    d = torch.empty_like(logits_lr)
    if gout.numel():
        scratch = torch.empty_like(gout)
        check(_lib.load().hiast_dinput_bwd(_ptr(logits_lr), int(bool(entropy)), _ptr(gout), _ptr(scratch), _ptr(d),
                                           B, C, h, w, H, W, _stream()), "hiast_dinput_bwd")
    else:
        d.zero_()
    return d


# ------------------------------------------------------------------------------- K5-K8 loss
REGION = {"ignored": 0, "confident": 1, "all": 2}


def _loss_args(logits_lr, teacher_lr, plbl, H, W):
    _req(logits_lr, torch.float32, 4, "logits_lr")
    B, C, h, w = logits_lr.shape
    if teacher_lr is not None:
        _req(teacher_lr, torch.float32, 4, "teacher_lr")
        assert teacher_lr.shape == logits_lr.shape
    if plbl.dtype not in (torch.uint8, torch.int64):
        raise TypeError("plbl must be uint8 or int64")
    _req(plbl, plbl.dtype, 3, "plbl")
    assert tuple(plbl.shape) == (B, H, W), (tuple(plbl.shape), (B, H, W))
    return B, C, h, w


def st_loss_workspace(B, C, h, w, H, W, device):
    n = _lib.load().hiast_st_loss_workspace_bytes(B, C, h, w, H, W)
    if n == 0:
        raise _lib.HiastLibraryError("hiast_st_loss: unsupported upsampling geometry %s -> %s" % ((h, w), (H, W)))
    return torch.empty((n + 7) // 8, dtype=torch.float64, device=device)


def st_loss_fwd(logits_lr, teacher_lr, plbl, H, W, region, workspace=None):
    """-> sums f64 [8] (device), see include/hiast_hip.h"""
    B, C, h, w = _loss_args(logits_lr, teacher_lr, plbl, H, W)
    ws = workspace if workspace is not None else st_loss_workspace(B, C, h, w, H, W, logits_lr.device)
    sums = torch.empty(8, dtype=torch.float64, device=logits_lr.device)
    check(_lib.load().hiast_st_loss_fwd(_ptr(logits_lr), _ptr(teacher_lr), _ptr(plbl),
                                        int(plbl.dtype == torch.int64), B, C, h, w, H, W, REGION[region],
                                        _ptr(sums), _ptr(ws), ws.numel() * 8, _stream()), "hiast_st_loss_fwd")
    return sums


def st_loss_bwd(logits_lr, teacher_lr, plbl, H, W, region, sums, coef, workspace=None):
    B, C, h, w = _loss_args(logits_lr, teacher_lr, plbl, H, W)
    _req(sums, torch.float64, 1, "sums")
    _req(coef, torch.float32, 1, "coef")
    assert sums.numel() == 8 and coef.numel() == 4
    ws = workspace if workspace is not None else st_loss_workspace(B, C, h, w, H, W, logits_lr.device)
    d = torch.empty_like(logits_lr)
    check(_lib.load().hiast_st_loss_bwd(_ptr(logits_lr), _ptr(teacher_lr), _ptr(plbl),
                                        int(plbl.dtype == torch.int64), B, C, h, w, H, W, REGION[region],
                                        _ptr(sums), _ptr(coef), _ptr(d), _ptr(ws), ws.numel() * 8, _stream()),
          "hiast_st_loss_bwd")
    return d


# ------------------------------------------------------------------------------- K1 ASPP
def _dil(dil):
    assert len(dil) == 4
    return (ctypes.c_int * 4)(*[int(d) for d in dil])


def aspp_pack_weights(weights, biases):
    assert len(weights) == 4 and len(biases) == 4
    Cout, Cin = weights[0].shape[:2]
    for wt, b in zip(weights, biases):
        _req(wt, torch.float32, 4, "aspp weight")
        _req(b, torch.float32, 1, "aspp bias")
        assert tuple(wt.shape) == (Cout, Cin, 3, 3) and b.numel() == Cout
    lib = _lib.load()
    n = lib.hiast_aspp_wpack_bytes(Cin, Cout)
    wpack = torch.empty(n // 4, dtype=torch.float32, device=weights[0].device)
    check(lib.hiast_aspp_pack_weights(*[_ptr(t) for t in weights], *[_ptr(t) for t in biases], Cin, Cout,
                                      _ptr(wpack), _stream()), "hiast_aspp_pack_weights")
    return wpack


def aspp_workspace(B, Cin, h, w, Cout, device):
    n = _lib.load().hiast_aspp_workspace_bytes(B, Cin, h, w, Cout)
    return torch.empty((n + 3) // 4, dtype=torch.float32, device=device)


def aspp_fwd(x, wpack, Cout, dil, workspace=None):
    _req(x, torch.float32, 4, "x")
    _req(wpack, torch.float32, 1, "wpack")
    B, Cin, h, w = x.shape
    lib = _lib.load()
    assert wpack.numel() * 4 == lib.hiast_aspp_wpack_bytes(Cin, Cout), "wpack does not match (Cin, Cout)"
    ws = workspace if workspace is not None else aspp_workspace(B, Cin, h, w, Cout, x.device)
    y = torch.empty((B, Cout, h, w), dtype=torch.float32, device=x.device)
    check(lib.hiast_aspp_fwd(_ptr(x), _ptr(wpack), _ptr(y), B, Cin, h, w, Cout, _dil(dil), _ptr(ws),
                             ws.numel() * 4, _stream()), "hiast_aspp_fwd")
    return y


def aspp_bwd_data(dy, wpack, Cin, dil):
    _req(dy, torch.float32, 4, "dy")
    _req(wpack, torch.float32, 1, "wpack")
    B, Cout, h, w = dy.shape
    lib = _lib.load()
    assert wpack.numel() * 4 == lib.hiast_aspp_wpack_bytes(Cin, Cout)
    dx = torch.empty((B, Cin, h, w), dtype=torch.float32, device=dy.device)
    check(lib.hiast_aspp_bwd_data(_ptr(dy), _ptr(wpack), _ptr(dx), B, Cin, h, w, Cout, _dil(dil), _stream()),
          "hiast_aspp_bwd_data")
    return dx


def aspp_bwd_weight(x, dy, dil, workspace=None):
    _req(x, torch.float32, 4, "x")
    _req(dy, torch.float32, 4, "dy")
    B, Cin, h, w = x.shape
    Cout = dy.shape[1]
    assert tuple(dy.shape) == (B, Cout, h, w)
    lib = _lib.load()
    ws = workspace if workspace is not None else aspp_workspace(B, Cin, h, w, Cout, x.device)
    dws = [torch.empty((Cout, Cin, 3, 3), dtype=torch.float32, device=x.device) for _ in range(4)]
    db = torch.empty((Cout,), dtype=torch.float32, device=x.device)
    check(lib.hiast_aspp_bwd_weight(_ptr(x), _ptr(dy), *[_ptr(t) for t in dws], _ptr(db), B, Cin, h, w, Cout,
                                    _dil(dil), _ptr(ws), ws.numel() * 4, _stream()), "hiast_aspp_bwd_weight")
    return dws, db


# ------------------------------------------------------------------------------- 16-bit operand formats
# HIAST_FMT_* of include/hiast_hip.h.  On the torch side a bf16-row tensor IS a torch.bfloat16 tensor and an fp16-row
# tensor a torch.float16 one; split planes are opaque bfloat16 tensors with twice the channels.
FMT_BF16, FMT_SPLIT_BF16, FMT_FP16 = 1, 2, 3
_H16 = (torch.bfloat16, torch.float16)


def fmt_of(t, planes=1):
    """operand format of a 16-bit activation / packed-weight tensor (planes = 2: split-bf16 planes)"""
    if int(planes) == 2:
        if t.dtype != torch.bfloat16:
            raise TypeError("split planes are bfloat16 tensors, got %s" % t.dtype)
        return FMT_SPLIT_BF16
    if t.dtype == torch.bfloat16:
        return FMT_BF16
    if t.dtype == torch.float16:
        return FMT_FP16
    raise TypeError("16-bit kernels take bfloat16 or float16 tensors, got %s" % t.dtype)


def fmt_dtype(fmt):
    return torch.float16 if int(fmt) == FMT_FP16 else torch.bfloat16


def _req16(t, ndim, name):
    """HIP tensor, bfloat16 or float16, contiguous"""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.HiastLibraryError("%s must be a CUDA(HIP) tensor: the HIP path has no CPU fallback" % name)
    if t.dtype not in _H16 or t.dim() != ndim or not t.is_contiguous():
        raise TypeError("%s must be a contiguous %d-d bfloat16 / float16 tensor (got %s, %s)" % (name, ndim, t.dtype, tuple(t.shape)))
    return t


# ------------------------------------------------------------------------------- K1b ASPP on NHWC (GEMM + shift-add)
def _nhwc_view(x):
    """logical [B,C,h,w] tensor whose memory is channels-last contiguous -> ([B,h,w,C] contiguous view, dtype code)"""
    if not isinstance(x, torch.Tensor) or not x.is_cuda:
        raise _lib.HiastLibraryError("x must be a CUDA(HIP) tensor: the HIP path has no CPU fallback")
    if x.dim() != 4:
        raise ValueError("x must be [B,C,h,w]")
    v = x.permute(0, 2, 3, 1)
    if not v.is_contiguous():
        raise ValueError("x must be channels-last contiguous")
    if x.dtype == torch.float32:
        return v, 0
    if x.dtype == torch.bfloat16:
        return v, FMT_BF16
    if x.dtype == torch.float16:
        return v, FMT_FP16
    raise TypeError("aspp_nhwc takes float32, bfloat16 or float16 activations, got %s" % x.dtype)


def aspp2_pack_weights(weights, biases, need_dgrad=True):
    """-> (wt [NP,Cin], wd [Cin,NP] or None, bias [Cout]) fp32"""
    assert len(weights) == 4 and len(biases) == 4
    Cout, Cin = weights[0].shape[:2]
    for wt_, b in zip(weights, biases):
        _req(wt_, torch.float32, 4, "aspp weight")
        _req(b, torch.float32, 1, "aspp bias")
        assert tuple(wt_.shape) == (Cout, Cin, 3, 3) and b.numel() == Cout
    lib = _lib.load()
    NP = lib.hiast_aspp2_np(Cout)
    if NP == 0:
        raise _lib.HiastLibraryError("hiast_aspp2: unsupported Cout %d" % Cout)
    dev = weights[0].device
    wt = torch.empty((NP, Cin), dtype=torch.float32, device=dev)
    wd = torch.empty((Cin, NP), dtype=torch.float32, device=dev) if need_dgrad else None
    bias = torch.empty((Cout,), dtype=torch.float32, device=dev)
    check(lib.hiast_aspp2_pack_weights(*[_ptr(t) for t in weights], *[_ptr(t) for t in biases], Cin, Cout, _ptr(wt),
                                       _ptr(wd), _ptr(bias), _stream()), "hiast_aspp2_pack_weights")
    return wt, wd, bias


def aspp2_workspace(B, Cin, h, w, Cout, backward, device):
    n = _lib.load().hiast_aspp2_workspace_bytes(B, Cin, h, w, Cout, int(bool(backward)))
    if n == 0:
        raise _lib.HiastLibraryError("hiast_aspp2: unsupported shape B=%d Cin=%d %dx%d Cout=%d" % (B, Cin, h, w, Cout))
    return torch.empty((n + 15) // 16 * 4, dtype=torch.float32, device=device)


def aspp2_fwd(x, wt, bias, dil, workspace=None, planes=None):
    """x: logical [B,Cin,h,w] with channels-last memory, fp32 (split-bf16 arithmetic) or bf16; or, with planes=2,
    a split-plane tensor [B,h,w,2*Cin] (K9c format).  wt: fp32 [NP,Cin] from aspp2_pack_weights.
    -> y [B,Cout,h,w] fp32 NCHW"""
    _req(wt, torch.float32, 2, "wt")
    _req(bias, torch.float32, 1, "bias")
    if planes == 2:
        _req(x, torch.bfloat16, 4, "x (split planes)")
        B, h, w, C2 = x.shape
        xv, dt, Cin = x, 2, C2 // 2
    else:
        xv, dt = _nhwc_view(x)
        B, h, w, Cin = xv.shape
    Cout = bias.numel()
    lib = _lib.load()
    assert tuple(wt.shape) == (lib.hiast_aspp2_np(Cout), Cin), "wt does not match (Cin, Cout)"
    wsrc = wt if dt == 0 else pack_conv_weight(wt, dt)       # bf16 rows / split planes: LDS-DMA kernel operand format
    ws = workspace if workspace is not None else aspp2_workspace(B, Cin, h, w, Cout, False, x.device)
    y = torch.empty((B, Cout, h, w), dtype=torch.float32, device=x.device)
    check(lib.hiast_aspp2_fwd(_ptr(xv), dt, _ptr(wsrc), _ptr(bias), _ptr(y), B, Cin, h, w, Cout, _dil(dil), _ptr(ws),
                              ws.numel() * 4, _stream()), "hiast_aspp2_fwd")
    return y


def aspp2_bwd(x, dy, wd, dil, want_dx=True, want_dw=True, workspace=None):
    """x bf16 / fp16 channels-last (logical [B,Cin,h,w]); dy [B,Cout,h,w] fp32; wd fp32 [Cin,NP] from aspp2_pack_weights
    -> (dx like x or None, [dW_i fp32 [Cout,Cin,3,3]] or None, db or None)"""
    xv, dt = _nhwc_view(x)
    if dt not in (FMT_BF16, FMT_FP16):
        raise TypeError("aspp2_bwd is the mixed-precision backward: x must be bfloat16 or float16")
    B, h, w, Cin = xv.shape
    _req(dy, torch.float32, 4, "dy")
    Cout = dy.shape[1]
    assert tuple(dy.shape) == (B, Cout, h, w)
    lib = _lib.load()
    NP = lib.hiast_aspp2_np(Cout)
    wdp = None
    if want_dx:
        _req(wd, torch.float32, 2, "wd")
        assert tuple(wd.shape) == (Cin, NP)
        wdp = pack_conv_weight(wd, dt)
    ws = workspace if workspace is not None else aspp2_workspace(B, Cin, h, w, Cout, True, x.device)
    dx = torch.empty((B, h, w, Cin), dtype=x.dtype, device=x.device) if want_dx else None
    dws = [torch.empty((Cout, Cin, 3, 3), dtype=torch.float32, device=x.device) for _ in range(4)] if want_dw else [None] * 4
    db = torch.empty((Cout,), dtype=torch.float32, device=x.device) if want_dw else None
    check(lib.hiast_aspp2_bwd(_ptr(xv), _ptr(dy), _ptr(wdp), _ptr(dx),
                              *[_ptr(t) for t in dws], _ptr(db), B, Cin, h, w, Cout, _dil(dil), dt, _ptr(ws),
                              ws.numel() * 4, _stream()), "hiast_aspp2_bwd")
    return (dx.permute(0, 3, 1, 2) if want_dx else None), (dws if want_dw else None), db


# ------------------------------------------------------------------------------- K11 EMA
class EmaPlan:
    """Device tables for one (ema parameters, student parameters) pairing; built once."""
    CHUNK = 65536

    def __init__(self, ema_tensors, src_tensors):
        assert len(ema_tensors) == len(src_tensors) and len(ema_tensors) > 0
        dev = ema_tensors[0].device
        recs = np.zeros((len(ema_tensors), 3), dtype=np.int64)
        ct, cs = [], []
        for i, (e, p) in enumerate(zip(ema_tensors, src_tensors)):
            _req(e, torch.float32, None, "ema tensor")
            _req(p, torch.float32, None, "student tensor")
            assert e.numel() == p.numel()
            recs[i] = (e.data_ptr(), p.data_ptr(), e.numel())
            for s in range(0, e.numel(), self.CHUNK):
                ct.append(i)
                cs.append(s)
        self.keep = (list(ema_tensors), list(src_tensors))
        self.table = torch.from_numpy(recs).to(dev)
        self.chunk_tensor = torch.tensor(ct, dtype=torch.int32, device=dev)
        self.chunk_start = torch.tensor(cs, dtype=torch.int64, device=dev)
        self.n_chunks = len(ct)
        self.ptrs = recs[:, :2].copy()

    def still_valid(self):
        return all(e.data_ptr() == a and p.data_ptr() == b
                   for (e, p, (a, b)) in zip(self.keep[0], self.keep[1], self.ptrs))

    def matches(self, ema_tensors, src_tensors):
        """the LIVE tensors still sit where the table points (a load_state_dict / .to() may have moved them)"""
        return len(ema_tensors) == len(self.ptrs) and all(
            e.data_ptr() == a and p.data_ptr() == b for (e, p, (a, b)) in zip(ema_tensors, src_tensors, self.ptrs))


def ema_update(plan, gamma):
    """ema = ema*gamma + p*(1-gamma), the scalars rounded to float32 as torch's tensor*float does"""
    check(_lib.load().hiast_ema_update(_ptr(plan.table), _ptr(plan.chunk_tensor), _ptr(plan.chunk_start),
                                       plan.n_chunks, float(np.float32(gamma)), float(np.float32(1 - gamma)),
                                       _stream()), "hiast_ema_update")


def normalize_u8(img_u8, mean, std):
    """uint8 [B,H,W,3] (HWC, as decoded) on the device -> float32 [B,3,H,W]: torchvision ToTensor + Normalize"""
    _req(img_u8, torch.uint8, 4, "img_u8")
    B, H, W, three = img_u8.shape
    assert three == 3 and len(mean) == 3 and len(std) == 3
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=img_u8.device)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s_ = (ctypes.c_float * 3)(*[float(v) for v in std])
    if B:
        check(_lib.load().hiast_normalize_u8(_ptr(img_u8), _ptr(out), B, H * W, m, s_, _stream()), "hiast_normalize_u8")
    return out


class CopyPlan:
    """device table for copying a fixed list of (small) tensors in one launch"""

    def __init__(self, dsts, srcs):
        assert len(dsts) == len(srcs) and len(dsts) > 0
        recs = np.zeros((len(dsts), 3), dtype=np.int64)
        for i, (d, s_) in enumerate(zip(dsts, srcs)):
            if not (d.is_cuda and s_.is_cuda and d.is_contiguous() and s_.is_contiguous() and d.dtype == s_.dtype
                    and d.numel() == s_.numel()):
                raise ValueError("multi_copy: tensor %d: contiguous HIP tensors of equal dtype and size required" % i)
            recs[i] = (d.data_ptr(), s_.data_ptr(), d.numel() * d.element_size())
        self.keep = (list(dsts), list(srcs))
        self.ptrs = recs[:, :2].copy()
        self.table = torch.from_numpy(recs).to(dsts[0].device)
        self.n = len(dsts)

    def still_valid(self):
        return all(d.data_ptr() == a and s_.data_ptr() == b for (d, s_, (a, b)) in zip(self.keep[0], self.keep[1], self.ptrs))

    def matches(self, dsts, srcs):
        return len(dsts) == self.n and all(d.data_ptr() == a and s_.data_ptr() == b
                                           for (d, s_, (a, b)) in zip(dsts, srcs, self.ptrs))


def multi_copy(plan):
    check(_lib.load().hiast_multi_copy(_ptr(plan.table), plan.n, _stream()), "hiast_multi_copy")


# ------------------------------------------------------------------------------- K13 Adam
class AdamPlan:
    """chunk tables for one list of parameter sizes (static); the pointer / lr records are rebuilt per step because
    gradient tensors are re-created by autograd after zero_grad(set_to_none=True)"""
    CHUNK = 65536
    REC = np.dtype([("p", np.int64), ("g", np.int64), ("m", np.int64), ("v", np.int64), ("n", np.int64),
                    ("lr", np.float32), ("bc1", np.float32), ("bc2s", np.float32), ("pad", np.float32)])

    def __init__(self, numels, device):
        ct, cs = [], []
        for i, n in enumerate(numels):
            for s0 in range(0, n, self.CHUNK):
                ct.append(i)
                cs.append(s0)
        self.numels = tuple(numels)
        self.chunk_tensor = torch.tensor(ct, dtype=torch.int32, device=device)
        self.chunk_start = torch.tensor(cs, dtype=torch.int64, device=device)
        self.n_chunks = len(ct)
        self.host = np.zeros(len(numels), dtype=self.REC)
        self.table = torch.empty(len(numels) * self.REC.itemsize, dtype=torch.uint8, device=device)


def adam_prepare(ctl, grad_scale=None, found_inf=None):
    """once per optimiser step, before its adam_step launches: the scaler's device scalars -> control block (skip flag,
    1 / scale, count of skipped steps).  ctl: float32 [8] device tensor (hiast_adam_ctl), zeroed when created"""
    _req(ctl, torch.float32, 1, "ctl")
    assert ctl.numel() == 8
    for t, nm in ((grad_scale, "grad_scale"), (found_inf, "found_inf")):
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.numel() == 1):
            raise ValueError("adam_prepare: %s must be a float32 device scalar" % nm)
    check(_lib.load().hiast_adam_prepare(_ptr(ctl), _ptr(grad_scale), _ptr(found_inf), _stream()), "hiast_adam_prepare")


def adam_step(plan, params, grads, exp_avgs, exp_avg_sqs, lrs, bc1s, bc2_sqrts, beta1, beta2, eps, weight_decay,
              ctl=None, steps=None):
    """one launch over all tensors; p/m/v updated in place.  With ctl (a control block adam_prepare has filled for this
    step) and steps (per-tensor count of ATTEMPTED steps, this one included) the bias corrections are formed on the device
    from steps[i] - skipped (bc1s / bc2_sqrts are ignored), the gradients are multiplied by 1 / grad_scale and the whole
    update is skipped on an overflow — no host synchronisation"""
    assert (ctl is None) == (steps is None)
    h = plan.host
    for i, (p, g, m, v) in enumerate(zip(params, grads, exp_avgs, exp_avg_sqs)):
        for t, nm in ((p, "param"), (g, "grad"), (m, "exp_avg"), (v, "exp_avg_sq")):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == plan.numels[i]):
                raise ValueError("adam_step: %s %d must be a contiguous float32 HIP tensor of %d elements"
                                 % (nm, i, plan.numels[i]))
        h[i] = (p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lrs[i], bc1s[i], bc2_sqrts[i],
                0.0 if steps is None else float(steps[i]))
    h2d_async(h.view(np.uint8).reshape(-1), plan.table.device, out=plan.table)
    check(_lib.load().hiast_adam_step(_ptr(plan.table), _ptr(plan.chunk_tensor), _ptr(plan.chunk_start), plan.n_chunks,
                                      float(beta1), float(beta2), float(eps), float(weight_decay), _ptr(ctl), _stream()),
          "hiast_adam_step")


# ------------------------------------------------------------------------------- K12 IoU
def confusion_hist(pred, target, K, inter=None, area_pred=None, area_tgt=None):
    _req(pred, torch.int64, None, "pred")
    _req(target, torch.int64, None, "target")
    assert pred.shape == target.shape
    dev = pred.device
    if inter is None:
        inter = torch.zeros(K, dtype=torch.int64, device=dev)
        area_pred = torch.zeros(K, dtype=torch.int64, device=dev)
        area_tgt = torch.zeros(K, dtype=torch.int64, device=dev)
    if pred.numel():
        check(_lib.load().hiast_confusion_hist(_ptr(pred), _ptr(target), pred.numel(), K, _ptr(inter),
                                               _ptr(area_pred), _ptr(area_tgt), _stream()),
              "hiast_confusion_hist")
    return inter, area_pred, area_tgt


# ------------------------------------------------------------------------------- K10 BN (+res) (+ReLU)
def _bn_dtype(t):
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.bfloat16:
        return 1
    if t.dtype == torch.float16:
        return 2
    raise TypeError("bn_act supports float32, bfloat16 and float16 activations, got %s" % t.dtype)


def _bn_dims(x):
    if not x.is_cuda:
        raise _lib.HiastLibraryError("bn_act runs on the HIP device only")
    assert x.dim() == 4 and x.is_contiguous()
    B, C, H, W = x.shape
    return B, C, H * W


def bn_stats(x):
    """-> part f64 [C, B, 2] = per-plane (Σx, Σx²)"""
    B, C, HW = _bn_dims(x)
    part = torch.empty((C, B, 2), dtype=torch.float64, device=x.device)
    check(_lib.load().hiast_bn_stats(_ptr(x), B, C, HW, _bn_dtype(x), _ptr(part), _stream()), "hiast_bn_stats")
    return part


def bn_act_apply(x, res, gamma, beta, running_mean, running_var, part, count, momentum, eps, relu):
    """part None -> inference with running statistics; else training with the given partial sums.
    -> (y, save_mean, save_invstd)  (the last two None in inference)"""
    B, C, HW = _bn_dims(x)
    y = torch.empty_like(x)
    if res is not None:
        assert res.shape == x.shape and res.dtype == x.dtype and res.is_contiguous()
    sm = si = None
    npart = 0
    if part is not None:
        assert part.dtype == torch.float64 and part.is_contiguous() and part.shape[0] == C and part.shape[2] == 2
        npart = part.shape[1]
        sm = torch.empty(C, dtype=torch.float32, device=x.device)
        si = torch.empty(C, dtype=torch.float32, device=x.device)
    check(_lib.load().hiast_bn_act_apply(_ptr(x), _ptr(res), _ptr(y), _ptr(gamma), _ptr(beta), _ptr(running_mean),
                                         _ptr(running_var), _ptr(part), npart, float(count), float(momentum),
                                         float(eps), int(bool(relu)), _ptr(sm), _ptr(si), B, C, HW, _bn_dtype(x),
                                         _stream()), "hiast_bn_act_apply")
    return y, sm, si


def bn_act_bwd_stats(dy, y, x, save_mean, save_invstd, relu):
    B, C, HW = _bn_dims(x)
    assert dy.shape == x.shape and dy.dtype == x.dtype and dy.is_contiguous()
    part = torch.empty((C, B, 2), dtype=torch.float64, device=x.device)
    check(_lib.load().hiast_bn_act_bwd_stats(_ptr(dy), _ptr(y), _ptr(x), _ptr(save_mean), _ptr(save_invstd),
                                             int(bool(relu)), B, C, HW, _bn_dtype(x), _ptr(part), _stream()),
          "hiast_bn_act_bwd_stats")
    return part


def bn_act_bwd_apply(dy, y, x, gamma, save_mean, save_invstd, part, count, relu, want_dres, want_dparam):
    B, C, HW = _bn_dims(x)
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    dg = torch.empty(C, dtype=torch.float32, device=x.device) if want_dparam else None
    db = torch.empty(C, dtype=torch.float32, device=x.device) if want_dparam else None
    check(_lib.load().hiast_bn_act_bwd_apply(_ptr(dy), _ptr(y), _ptr(x), _ptr(gamma), _ptr(save_mean),
                                             _ptr(save_invstd), _ptr(part), part.shape[1], float(count),
                                             int(bool(relu)), _ptr(dx), _ptr(dres), _ptr(dg), _ptr(db), B, C, HW,
                                             _bn_dtype(x), _stream()), "hiast_bn_act_bwd_apply")
    return dx, dres, dg, db


# ------------------------------------------------------------------------------- K9 NHWC inference convs
def _act_dtype(t):
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.bfloat16:
        return 1
    raise TypeError("NHWC conv kernels take float32 or bfloat16 activations, got %s" % t.dtype)


def _act_dtype16(t):
    """0 fp32 | 1 bf16 | 2 fp16 (hiast_stem_tail's input code)"""
    return 2 if t.dtype == torch.float16 else _act_dtype(t)


def _bn_params(bn):
    return (_ptr(bn.weight), _ptr(bn.bias), _ptr(bn.running_mean), _ptr(bn.running_var), float(bn.eps))


def bn_act_nhwc_infer(x2d, bn, relu=True):
    dt = _act_dtype(x2d)
    _req(x2d, x2d.dtype, 2, "x2d")
    M, C = x2d.shape
    y = torch.empty_like(x2d)
    g, b, mu, var, eps = _bn_params(bn)
    check(_lib.load().hiast_bn_act_nhwc_infer(_ptr(x2d), _ptr(y), g, b, mu, var, eps, int(bool(relu)), M, C, dt,
                                              _stream()), "hiast_bn_act_nhwc_infer")
    return y


def maxpool3x3s2_cl_fwd(x):
    """K18: x logical [B,C,H,W] bf16 / fp16 with channels-last memory -> (y like x at [B,C,Ho,Wo], idx uint8 [B,Ho,Wo,C])"""
    if not (x.is_cuda and x.dtype in _H16 and x.dim() == 4 and x.permute(0, 2, 3, 1).is_contiguous()):
        raise ValueError("maxpool3x3s2_cl: x must be a channels-last bfloat16 / float16 HIP tensor [B,C,H,W]")
    B, C, H, W = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((B, C, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    idx = torch.empty((B, Ho, Wo, C), dtype=torch.uint8, device=x.device)
    check(_lib.load().hiast_maxpool3x3s2_nhwc_fwd(_ptr(x), _ptr(y), _ptr(idx), B, H, W, C, fmt_of(x), _stream()),
          "hiast_maxpool3x3s2_nhwc_fwd")
    return y, idx


def maxpool3x3s2_cl_bwd(dy, idx, H, W):
    """dy logical [B,C,Ho,Wo] bf16 / fp16 channels-last, idx from the forward -> dx [B,C,H,W] like dy, channels-last"""
    if not (dy.is_cuda and dy.dtype in _H16 and dy.dim() == 4 and dy.permute(0, 2, 3, 1).is_contiguous()):
        raise ValueError("maxpool3x3s2_cl: dy must be a channels-last bfloat16 / float16 HIP tensor [B,C,Ho,Wo]")
    B, C, Ho, Wo = dy.shape
    assert tuple(idx.shape) == (B, Ho, Wo, C) and idx.dtype == torch.uint8
    dx = torch.empty((B, C, H, W), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
    check(_lib.load().hiast_maxpool3x3s2_nhwc_bwd(_ptr(dy), _ptr(idx), _ptr(dx), B, H, W, C, fmt_of(dy), _stream()),
          "hiast_maxpool3x3s2_nhwc_bwd")
    return dx


def stem_tail(x, bn, fmt):
    """K9f: bn (eval) -> ReLU -> MaxPool2d(3, 2, 1) of the stem convolution's output in one pass.
    x: logical [B,C,H,W] with channels-last memory, fp32 / bf16 / fp16 -> [B,Ho,Wo,planes*C] in operand format `fmt`
    (FMT_BF16 | FMT_SPLIT_BF16 | FMT_FP16)"""
    dt = _act_dtype16(x)
    fmt = int(fmt)
    B, C, H, W = x.shape
    assert x.permute(0, 2, 3, 1).is_contiguous(), "stem_tail: x must be channels-last"
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((B, Ho, Wo, (2 if fmt == FMT_SPLIT_BF16 else 1) * C), dtype=fmt_dtype(fmt), device=x.device)
    g, b, mu, var, eps = _bn_params(bn)
    check(_lib.load().hiast_stem_tail(_ptr(x), dt, g, b, mu, var, eps, _ptr(out), fmt, B, H, W, C, _stream()),
          "hiast_stem_tail")
    return out


def stem_eval_supported(conv, bn, pool):
    """the module triple hiast_stem_eval replaces: Conv2d(3, 64, 7, 2, 3, bias=False) -> BatchNorm2d(64) (eval) -> ReLU ->
    MaxPool2d(3, 2, 1)"""
    import os
    return (os.environ.get("HIAST_NO_STEM_FUSED", "0") != "1"
            and (conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, conv.dilation, conv.groups)
            == (3, 64, (7, 7), (2, 2), (3, 3), (1, 1), 1) and conv.bias is None and bn.num_features == 64 and not bn.training
            and (pool.kernel_size, pool.stride, pool.padding, pool.dilation, pool.ceil_mode) == (3, 2, 1, 1, False))


def stem_eval(x, weight, bn, fmt):
    """K9j: conv 7x7 s2 -> bn (eval) -> ReLU -> MaxPool2d(3, 2, 1) in one kernel.
    x fp32 [B,3,H,W] -> [B,Hp,Wp,planes*64] in operand format `fmt` (FMT_BF16 | FMT_SPLIT_BF16 | FMT_FP16)"""
    x = x.contiguous()                    # NCHW (a channels-last or sliced batch is copied: 6 MB per image)
    _req(x, torch.float32, 4, "x")
    _req(weight, torch.float32, 4, "stem weight")
    assert tuple(weight.shape) == (64, 3, 7, 7) and x.shape[1] == 3
    fmt = int(fmt)
    B, _, H, W = x.shape
    Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1
    out = torch.empty((B, Hp, Wp, (2 if fmt == FMT_SPLIT_BF16 else 1) * 64), dtype=fmt_dtype(fmt), device=x.device)
    g, b, mu, var, eps = _bn_params(bn)
    check(_lib.load().hiast_stem_eval(_ptr(x), _ptr(weight), g, b, mu, var, eps, _ptr(out), fmt, B, H, W, _stream()),
          "hiast_stem_eval")
    return out


def stem_train_supported(conv):
    """the module hiast_stem_train_fwd / hiast_stem_wgrad replace: Conv2d(3, 64, 7, 2, 3, bias=False)"""
    return (not _SW.on("HIAST_LIB_STEM")
            and (conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, conv.dilation, conv.groups)
            == (3, 64, (7, 7), (2, 2), (3, 3), (1, 1), 1) and conv.bias is None)


def stem_train_shape_ok(x):
    """the K9k stem kernels take this batch (their 31-bit addressing: B*Hc*Wc*64 and B*3*H*W below 2^31) — otherwise the caller
    keeps the library stem (the image gradient, which K9k never computes, is the caller's other reason to)"""
    B, _, H, W = x.shape
    return x.dtype == torch.float32 and _lib.load().hiast_stem_train_blocks(int(B), int(H), int(W)) > 0


def stem_train_fwd(x, weight, fmt):
    """K9k: the stem convolution of the training forward.  x fp32 [B,3,H,W] (NCHW), weight fp32 [64,3,7,7] ->
    (y [B,Hc,Wc,64] 16-bit channels-last rows in format fmt, partial fp32 [blocks,64,2]: Σy, Σy² of the stored values)"""
    _req(x, torch.float32, 4, "x")
    _req(weight, torch.float32, 4, "stem weight")
    assert tuple(weight.shape) == (64, 3, 7, 7) and x.shape[1] == 3
    fmt = int(fmt)
    B, _, H, W = x.shape
    Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    lib = _lib.load()
    nblk = lib.hiast_stem_train_blocks(B, H, W)
    if nblk <= 0:
        raise _lib.HiastLibraryError("hiast_stem_train_fwd: unsupported geometry B=%d H=%d W=%d" % (B, H, W))
    y = torch.empty((B, Hc, Wc, 64), dtype=fmt_dtype(fmt), device=x.device)
    partial = torch.empty((nblk, 64, 2), dtype=torch.float32, device=x.device)
    check(lib.hiast_stem_train_fwd(_ptr(x), _ptr(weight), _ptr(y), _ptr(partial), fmt, B, H, W, _stream()),
          "hiast_stem_train_fwd")
    return y, partial


_stem_ws = {}


def stem_wgrad(x, dy):
    """K9k: dW fp32 [64,3,7,7] of the stem convolution from x fp32 [B,3,H,W] (NCHW) and dy [B,Hc,Wc,64] 16-bit rows"""
    _req(x, torch.float32, 4, "x")
    _req16(dy, 4, "dy")
    B, _, H, W = x.shape
    assert tuple(dy.shape) == (B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 64)
    lib = _lib.load()
    need = lib.hiast_stem_wgrad_workspace_bytes(B, H, W)
    if need == 0:
        raise _lib.HiastLibraryError("hiast_stem_wgrad: unsupported geometry B=%d H=%d W=%d" % (B, H, W))
    key = (x.device, torch.cuda.current_stream(x.device).cuda_stream)
    ws = _stem_ws.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device)
        _stem_ws[key] = ws
    dw = torch.empty((64, 3, 7, 7), dtype=torch.float32, device=x.device)
    check(lib.hiast_stem_wgrad(_ptr(x), _ptr(dy), _ptr(dw), fmt_of(dy), B, H, W, _ptr(ws), ws.numel() * 4, _stream()),
          "hiast_stem_wgrad")
    return dw


# ------------------------------------------------------------------------------- K9c LDS-DMA implicit GEMM on split planes
# A "split-plane" activation is an opaque bf16 tensor [B,H,W,2*C] holding hi = bf16(v) and lo = bf16(v - hi) of an
# fp32-class value (layout inside the last axis: include/hiast_hip.h, K9c); plain bf16 activations are [B,H,W,C].
def split_planes(x2d):
    """fp32 [M,C] (channels-last rows) -> bf16 [M,2*C] split planes"""
    _req(x2d, torch.float32, 2, "x2d")
    M, C = x2d.shape
    p = torch.empty((M, 2 * C), dtype=torch.bfloat16, device=x2d.device)
    check(_lib.load().hiast_split_planes(_ptr(x2d), _ptr(p), M, C, 0, _stream()), "hiast_split_planes")
    return p


def merge_planes(p):
    """bf16 [M,2*C] split planes -> fp32 [M,C] (hi + lo, exact)"""
    _req(p, torch.bfloat16, 2, "planes")
    M, C2 = p.shape
    x = torch.empty((M, C2 // 2), dtype=torch.float32, device=p.device)
    check(_lib.load().hiast_split_planes(_ptr(x), _ptr(p), M, C2 // 2, 1, _stream()), "hiast_split_planes")
    return x


def pack_conv_weight(weight, fmt, transpose=False, both=False):
    """fp32 conv weight [N,K,kh,kw] (torch layout) -> packed [N, kh*kw, planes*K] in operand format `fmt` (FMT_BF16 = 1,
    FMT_SPLIT_BF16 = 2: two bf16 planes, FMT_FP16 = 3); transpose=True packs the adjoint (data-gradient) convolution's
    weight [K, kh*kw (flipped), planes*N]; both=True -> (forward, adjoint) in one launch"""
    fmt = int(fmt)
    planes, h16 = (2 if fmt == FMT_SPLIT_BF16 else 1), fmt_dtype(fmt)
    w = weight.detach()
    if w.dim() == 2:
        w = w.reshape(w.shape[0], w.shape[1], 1, 1)
    _req(w, torch.float32, 4, "weight")
    N, K_, kh, kw = w.shape
    fwd_shape, adj_shape = (N, kh * kw, planes * K_), (K_, kh * kw, planes * N)
    wpt = None
    if both:
        wp = torch.empty(fwd_shape, dtype=h16, device=w.device)
        wpt = torch.empty(adj_shape, dtype=h16, device=w.device)
        mode = 2
    else:
        wp = torch.empty(adj_shape if transpose else fwd_shape, dtype=h16, device=w.device)
        mode = 1 if transpose else 0
    check(_lib.load().hiast_pack_conv_weight(_ptr(w), N, K_, kh * kw, fmt, mode, _ptr(wp), _ptr(wpt), _stream()),
          "hiast_pack_conv_weight")
    return (wp, wpt) if both else wp


class PackPlan:
    """persistent packed copies (forward and, optionally, adjoint) of a fixed list of conv weights, refreshed by ONE
    launch (hiast_pack_conv_weight_multi) — e.g. the 104 convolutions of a trunk after each optimiser / EMA step"""
    CHUNK = 65536
    REC = np.dtype([("w", np.int64), ("wp", np.int64), ("wpt", np.int64), ("N", np.int32), ("K", np.int32),
                    ("taps", np.int32), ("planes", np.int32), ("mode", np.int32), ("pad", np.int32)])

    def __init__(self, weights, fmt, adjoint):
        """weights: list of fp32 [N,K,kh,kw] parameters; fmt: operand format (FMT_*); adjoint: list of bool (also keep the
        data-gradient form)"""
        assert len(weights) > 0 and len(weights) == len(adjoint)
        dev = weights[0].device
        fmt = int(fmt)
        planes, h16 = (2 if fmt == FMT_SPLIT_BF16 else 1), fmt_dtype(fmt)
        self.weights, self.fmt = list(weights), fmt
        self.wp, self.wpt = [], []
        host = np.zeros(len(weights), dtype=self.REC)
        lists = {}                     # taps -> ([record index per block], [tile index per block]): one launch per kernel size
        for i, (w, adj) in enumerate(zip(weights, adjoint)):
            _req(w.detach(), torch.float32, 4, "weight")
            N, K_, kh, kw = w.shape
            taps = kh * kw
            wp = torch.empty((N, taps, planes * K_), dtype=h16, device=dev)
            wpt = torch.empty((K_, taps, planes * N), dtype=h16, device=dev) if adj else None
            self.wp.append(wp)
            self.wpt.append(wpt)
            host[i] = (w.data_ptr(), wp.data_ptr(), wpt.data_ptr() if adj else 0, N, K_, taps, fmt, 2 if adj else 0, 0)
            if N % 64 or K_ % 64 or taps > 9:
                raise ValueError("PackPlan: channel counts must be multiples of 64 (got %d x %d)" % (N, K_))
            ct, cs = lists.setdefault(taps, ([], []))
            for tile in range((N // 64) * (K_ // 64)):      # one block per 64 x 64 (n, k) tile, all taps
                ct.append(i)
                cs.append(tile)
        self.ptrs = [w.data_ptr() for w in weights]
        self.table = torch.from_numpy(host.view(np.uint8).reshape(-1).copy()).to(dev)
        self.launches = [(taps, torch.tensor(ct, dtype=torch.int32, device=dev), torch.tensor(cs, dtype=torch.int64, device=dev),
                          len(ct)) for taps, (ct, cs) in sorted(lists.items())]
        self.versions = None

    def still_valid(self):
        return all(w.data_ptr() == p for w, p in zip(self.weights, self.ptrs))

    def refresh(self):
        """re-pack if any weight has been modified since the last pack; -> True if a launch was made"""
        v = [w._version for w in self.weights]
        if v == self.versions:
            return False
        for taps, ct, cs, n in self.launches:        # the 1x1 and the 3x3 weights: LDS tile sized per kernel size
            check(_lib.load().hiast_pack_conv_weight_multi(_ptr(self.table), _ptr(ct), _ptr(cs), n, taps, _stream()),
                  "hiast_pack_conv_weight_multi")
        self.versions = v
        return True


def igemm_bn_act(x, wp, planes, bn, res, relu, stride=1, dil=1, out_f32=False, want_stats=False, res_gate=None):
    """x: [B,H,W,planes*Cin] (planes = 2: split-bf16 planes | 1: plain bf16 or fp16 rows — the tensor's dtype says which);
    wp from pack_conv_weight (same format); bn: BatchNorm2d in eval mode or None (plain GEMM); res: like the output or None
    -> y like x [B,Ho,Wo,planes*Cout], or fp32 [B,Ho,Wo,Cout] if out_f32;
    want_stats (planes = 1): -> (y, partial fp32 [rows, Cout, 2]) per-block Σy, Σy² for a following BatchNorm"""
    _req16(x, 4, "x")
    PL = 2 if int(planes) == 2 else 1
    fmt = fmt_of(x, PL)
    _req(wp, x.dtype, 3, "wp")
    B, H, W, CC = x.shape
    Cin = CC // PL
    N, taps, KK = wp.shape
    assert PL in (1, 2) and KK == PL * Cin and CC == PL * Cin and taps in (1, 9), (tuple(x.shape), tuple(wp.shape), PL)
    if taps == 1:
        assert stride == 1
        Ho, Wo = H, W
    else:
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if out_f32:
        y = torch.empty((B, Ho, Wo, N), dtype=torch.float32, device=x.device)
        assert res is None
    else:
        y = torch.empty((B, Ho, Wo, PL * N), dtype=x.dtype, device=x.device)
    if res is not None:
        _req(res, x.dtype, 4, "res")
        assert tuple(res.shape) == tuple(y.shape)
    gate_mask = 0
    if res_gate is not None:        # residual added only where the gate is open (planes = 1)
        assert PL == 1 and res is not None
        if res_gate.dtype == torch.uint8:        # bit mask [M, Cout/8] from bn_nhwc_apply(..., want_mask=True)
            _req(res_gate, torch.uint8, 2, "res_gate")
            assert tuple(res_gate.shape) == (B * Ho * Wo, N // 8)
            gate_mask = 1
        else:                                    # values: gate = res_gate > 0
            _req(res_gate, x.dtype, 4, "res_gate")
            assert tuple(res_gate.shape) == tuple(res.shape)
    if bn is not None:
        g, b, mu, var, eps = _bn_params(bn)
    else:
        g = b = mu = var = ctypes.c_void_p(0)
        eps = 0.0
    partial = None
    rows = 0
    if want_stats:
        assert PL == 1 and not out_f32
        rows = _lib.load().hiast_igemm_stats_rows(B * Ho * Wo, Cin, N, taps, fmt)    # one row per block of the kernel
        partial = torch.empty((rows, N, 2), dtype=torch.float32, device=x.device)
    # (the launch checks `rows` against the tile form it takes: HIAST_E_ARG instead of an overrun if the two ever disagree)
    check(_lib.load().hiast_igemm_bn_act(_ptr(x), _ptr(wp), g, b, mu, var, eps, _ptr(res), int(bool(relu)), _ptr(y),
                                         B, H, W, Cin, N, taps, int(stride), int(dil), fmt, int(bool(out_f32)),
                                         _ptr(partial), rows, _ptr(res_gate), gate_mask, _stream()), "hiast_igemm_bn_act")
    return (y, partial) if want_stats else y


def igemm_dgrad_bn_stats(dy, wpt, dil, bn_x, gamma, beta, save_mean, save_invstd):
    """data gradient of a stride-1 trunk convolution whose input was A = relu(bn(bn_x)):
    dy bf16 [B,H,W,Cin'], wpt = adjoint packed weight [Cout', taps, Cin'] -> (dA bf16 [B,H,W,Cout'],
    partial fp32 [rows, Cout', 2] = per-block (Σg, Σ g*xhat) of that BatchNorm's backward; bn_nhwc_stats_from_partial
    reduces them to the sums bn_nhwc_bwd_apply takes).  bn_x: bf16 [B,H,W,Cout'] (channels-last rows)."""
    _req16(dy, 4, "dy")
    _req(wpt, dy.dtype, 3, "wpt")
    _req(bn_x, dy.dtype, 4, "bn_x")
    B, H, W, Cin = dy.shape
    N, taps, KK = wpt.shape
    assert KK == Cin and taps in (1, 9) and tuple(bn_x.shape) == (B, H, W, N), (tuple(dy.shape), tuple(wpt.shape), tuple(bn_x.shape))
    _req(save_mean, torch.float32, 1, "save_mean")
    _req(save_invstd, torch.float32, 1, "save_invstd")
    lib = _lib.load()
    da = torch.empty((B, H, W, N), dtype=dy.dtype, device=dy.device)
    rows = lib.hiast_igemm_dgrad_bn_stats_rows(B * H * W, Cin, N, taps)     # one row per block of the tile form the launch takes
    partial = torch.empty((rows, N, 2), dtype=torch.float32, device=dy.device)
    check(lib.hiast_igemm_dgrad_bn_stats(_ptr(dy), _ptr(wpt), _ptr(da), B, H, W, Cin, N, taps, int(dil), _ptr(bn_x),
                                         _ptr(gamma), _ptr(beta), _ptr(save_mean), _ptr(save_invstd), _ptr(partial),
                                         rows, fmt_of(dy), _stream()), "hiast_igemm_dgrad_bn_stats")
    return da, partial


def xconv_dgrad_gated_bn_stats_ok(M, K, N):
    """shapes hiast_xconv_dgrad_gated_bn_stats takes (conv1 of a layer3 identity block)"""
    return _lib.load().hiast_xconv_dgrad_gated_bn_stats_rows(int(M), int(K), int(N)) > 0


def xconv_dgrad_gated_bn_stats(dy, wpt, res, res_gate, bn_x, bn_mask, save_mean, save_invstd):
    """conv1's data gradient of an identity bottleneck + the backward sums of the previous block's bn3 (K9e'):
    dy [B,H,W,K] (K = 256), wpt adjoint packed [N,1,K], res [B,H,W,N] + res_gate u8 [M,N/8] (this block's output gradient and
    ReLU gate bits), bn_x [B,H,W,N] + bn_mask u8 [M,N/8] + save_mean / save_invstd [N] (the previous block's bn3)
    -> (dx [B,H,W,N], partial fp32 [rows,N,2])"""
    _req16(dy, 4, "dy")
    _req(wpt, dy.dtype, 3, "wpt")
    _req(res, dy.dtype, 4, "res")
    _req(bn_x, dy.dtype, 4, "bn_x")
    B, H, W, Kc = dy.shape
    N, taps, KK = wpt.shape
    M = B * H * W
    assert taps == 1 and KK == Kc and tuple(res.shape) == (B, H, W, N) and tuple(bn_x.shape) == (B, H, W, N)
    for t, nm in ((res_gate, "res_gate"), (bn_mask, "bn_mask")):
        _req(t, torch.uint8, 2, nm)
        assert tuple(t.shape) == (M, N // 8), nm
    _req(save_mean, torch.float32, 1, "save_mean")
    _req(save_invstd, torch.float32, 1, "save_invstd")
    lib = _lib.load()
    rows = lib.hiast_xconv_dgrad_gated_bn_stats_rows(M, Kc, N)
    if rows <= 0:
        raise _lib.HiastLibraryError("hiast_xconv_dgrad_gated_bn_stats: unsupported shape M=%d K=%d N=%d" % (M, Kc, N))
    dx = torch.empty((B, H, W, N), dtype=dy.dtype, device=dy.device)
    partial = torch.empty((rows, N, 2), dtype=torch.float32, device=dy.device)
    check(lib.hiast_xconv_dgrad_gated_bn_stats(_ptr(dy), _ptr(wpt), _ptr(res), _ptr(res_gate), _ptr(bn_x), _ptr(bn_mask),
                                               _ptr(save_mean), _ptr(save_invstd), _ptr(dx), _ptr(partial), M, Kc, N,
                                               fmt_of(dy), _stream()), "hiast_xconv_dgrad_gated_bn_stats")
    return dx, partial


def igemm_dgrad_s2(dy, wpt, H, W):
    """data gradient of a 3x3 / stride-2 / padding-1 trunk convolution: dy [B,(H-1)//2+1,(W-1)//2+1,Cout] 16-bit rows,
    wpt = adjoint packed weight [Cin, 9, Cout] -> dx [B,H,W,Cin] (hiast_igemm_dgrad_s2: the tile kernel's transposed form)"""
    _req16(dy, 4, "dy")
    _req(wpt, dy.dtype, 3, "wpt")
    B, Hs, Ws, Cout = dy.shape
    N, taps, KK = wpt.shape
    assert taps == 9 and KK == Cout and (Hs, Ws) == ((H - 1) // 2 + 1, (W - 1) // 2 + 1), (tuple(dy.shape), tuple(wpt.shape), H, W)
    dx = torch.empty((B, H, W, N), dtype=dy.dtype, device=dy.device)
    check(_lib.load().hiast_igemm_dgrad_s2(_ptr(dy), _ptr(wpt), _ptr(dx), B, H, W, N, Cout, fmt_of(dy), _stream()),
          "hiast_igemm_dgrad_s2")
    return dx


# ------------------------------------------------------------------------------- K10b BN (train) on channels-last bf16
def _bnh_view(t, name):
    """logical [B,C,H,W] bf16 tensor with channels-last memory -> ([M,C] view, M, C)"""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.HiastLibraryError("%s must be a CUDA(HIP) tensor: the HIP path has no CPU fallback" % name)
    if t.dtype not in _H16 or t.dim() != 4:
        raise TypeError("%s must be a 4-d bfloat16 / float16 tensor" % name)
    v = t.permute(0, 2, 3, 1)
    if not v.is_contiguous():
        raise ValueError("%s must be channels-last contiguous" % name)
    B, H, W, C = v.shape
    return v, B * H * W, C


_bnh_ws = {}


def _bnh_workspace(C, device):
    key = (C, device)
    ws = _bnh_ws.get(key)
    if ws is None:
        n = _lib.load().hiast_bn_nhwc_workspace_bytes(C)
        ws = torch.empty(n // 4, dtype=torch.float32, device=device)
        _bnh_ws[key] = ws
    return ws


def bn_nhwc_supported(C):
    return 8 <= C <= 2048 and (C & (C - 1)) == 0


def bn_nhwc_stats(x):
    """-> sums f64 [C,2] = (Σx, Σx²)"""
    xv, M, C = _bnh_view(x, "x")
    sums = torch.empty((C, 2), dtype=torch.float64, device=x.device)
    ws = _bnh_workspace(C, x.device)
    check(_lib.load().hiast_bn_nhwc_stats(_ptr(xv), M, C, _ptr(sums), _ptr(ws), ws.numel() * 4, fmt_of(x), _stream()),
          "hiast_bn_nhwc_stats")
    return sums


def bn_nhwc_stats_from_partial(partial):
    """per-block partial sums [nblk,C,2] fp32 (igemm_bn_act(..., want_stats=True)) -> sums f64 [C,2]"""
    _req(partial, torch.float32, 3, "partial")
    nblk, C, two = partial.shape
    assert two == 2
    sums = torch.empty((C, 2), dtype=torch.float64, device=partial.device)
    check(_lib.load().hiast_bn_nhwc_stats_from_partial(_ptr(partial), nblk, C, _ptr(sums), _stream()),
          "hiast_bn_nhwc_stats_from_partial")
    return sums


def bn_nhwc_apply(x, res, gamma, beta, running_mean, running_var, sums, count, momentum, eps, relu, want_mask=False):
    """-> (y, save_mean, save_invstd[, mask u8 [M, C/8]: bits y > 0])"""
    xv, M, C = _bnh_view(x, "x")
    y = torch.empty_like(x, memory_format=torch.channels_last)
    mask = torch.empty((M, C // 8), dtype=torch.uint8, device=x.device) if want_mask else None
    rv = None
    if res is not None:
        rv, M2, C2 = _bnh_view(res, "res")
        assert (M2, C2) == (M, C)
    assert sums.dtype == torch.float64 and tuple(sums.shape) == (C, 2) and sums.is_contiguous()
    sm = torch.empty(C, dtype=torch.float32, device=x.device)
    si = torch.empty(C, dtype=torch.float32, device=x.device)
    check(_lib.load().hiast_bn_nhwc_apply(_ptr(xv), _ptr(rv), _ptr(y), _ptr(gamma), _ptr(beta), _ptr(running_mean),
                                          _ptr(running_var), _ptr(sums), float(count), float(momentum), float(eps),
                                          int(bool(relu)), _ptr(sm), _ptr(si), M, C, _ptr(mask), fmt_of(x), _stream()),
          "hiast_bn_nhwc_apply")
    return (y, sm, si, mask) if want_mask else (y, sm, si)


def bn_nhwc_apply_partial(x, res, gamma, beta, running_mean, running_var, partial, count, momentum, eps, relu,
                          want_mask=False):
    """single-rank forward from per-block partial sums [nblk,C,2] (igemm_bn_act(..., want_stats=True))"""
    xv, M, C = _bnh_view(x, "x")
    mask = torch.empty((M, C // 8), dtype=torch.uint8, device=x.device) if want_mask else None
    _req(partial, torch.float32, 3, "partial")
    assert partial.shape[1] == C and partial.shape[2] == 2
    y = torch.empty_like(x, memory_format=torch.channels_last)
    rv = _bnh_view(res, "res")[0] if res is not None else None
    sm = torch.empty(C, dtype=torch.float32, device=x.device)
    si = torch.empty(C, dtype=torch.float32, device=x.device)
    check(_lib.load().hiast_bn_nhwc_apply_partial(_ptr(xv), _ptr(rv), _ptr(y), _ptr(gamma), _ptr(beta), _ptr(running_mean),
                                                  _ptr(running_var), _ptr(partial), partial.shape[0], float(count),
                                                  float(momentum), float(eps), int(bool(relu)), _ptr(sm), _ptr(si), M, C,
                                                  _ptr(mask), fmt_of(x), _stream()), "hiast_bn_nhwc_apply_partial")
    return (y, sm, si, mask) if want_mask else (y, sm, si)


def bn_nhwc_bwd_stats(dy, y, x, gamma, beta, save_mean, save_invstd, gate):
    """gate: 0 = no ReLU, 1 = y > 0 (reads y), 2 = recomputed from x (forward without residual; y unused),
    3 = y is the bit mask u8 [M, C/8] the forward wrote"""
    xv, M, C = _bnh_view(x, "x")
    dv, M2, C2 = _bnh_view(dy, "dy")
    assert (M2, C2) == (M, C)
    yv = _bnh_view(y, "y")[0] if gate == 1 else (_req(y, torch.uint8, 2, "mask") if gate == 3 else None)
    sums = torch.empty((C, 2), dtype=torch.float64, device=x.device)
    ws = _bnh_workspace(C, x.device)
    check(_lib.load().hiast_bn_nhwc_bwd_stats(_ptr(dv), _ptr(yv), _ptr(xv), _ptr(gamma), _ptr(beta), _ptr(save_mean),
                                              _ptr(save_invstd), int(gate), M, C, _ptr(sums), _ptr(ws), ws.numel() * 4,
                                              fmt_of(x), _stream()), "hiast_bn_nhwc_bwd_stats")
    return sums


def bn_nhwc_bwd_apply(dy, y, x, gamma, beta, save_mean, save_invstd, sums, count, gate, want_dres, want_dparam):
    xv, M, C = _bnh_view(x, "x")
    dv = _bnh_view(dy, "dy")[0]
    yv = _bnh_view(y, "y")[0] if gate == 1 else (_req(y, torch.uint8, 2, "mask") if gate == 3 else None)
    dx = torch.empty_like(x, memory_format=torch.channels_last)
    dres = torch.empty_like(x, memory_format=torch.channels_last) if want_dres else None
    dg = torch.empty(C, dtype=torch.float32, device=x.device) if want_dparam else None
    db = torch.empty(C, dtype=torch.float32, device=x.device) if want_dparam else None
    check(_lib.load().hiast_bn_nhwc_bwd_apply(_ptr(dv), _ptr(yv), _ptr(xv), _ptr(gamma), _ptr(beta), _ptr(save_mean),
                                              _ptr(save_invstd), _ptr(sums), float(count), int(gate), _ptr(dx),
                                              _ptr(dres), _ptr(dg), _ptr(db), M, C, fmt_of(x), _stream()),
          "hiast_bn_nhwc_bwd_apply")
    return dx, dres, dg, db


# ------------------------------------------------------------------------------- K9d conv weight gradient (NHWC bf16)
_wgrad_ws = {}


def conv_wgrad_supported(Cin, Cout, k, stride):
    """shapes hiast_conv_wgrad_nhwc accepts"""
    return Cin % 256 == 0 and Cout % 256 == 0 and k in (1, 3) and (k == 3 or stride == 1)


def conv_wgrad_preferred(Cin, Cout, k, stride):
    """shapes on which it beats (3x3 at 512 channels: ties) the library on MI355X (measured, tools/bench_kernels.py
    wgrad, B=8): 1x1 layer3 0.062 vs 0.112 ms, layer4 0.225 vs 0.255 ms; 3x3 layer3 0.153 vs 0.165 ms, layer4 0.490 vs
    0.488 ms — and the library's result still needs a cast + layout kernel.  (The 3x3 form used to trail at 0.30 ms:
    two rounds of blocks doubled the partial-tile traffic and the nine tap blocks of a pixel range were dealt to eight
    different L2s.)  HIAST_LIB_WGRAD3=1 sends the 3x3 shapes to the library."""
    import os
    if not conv_wgrad_supported(Cin, Cout, k, stride):
        return False
    return k == 1 or os.environ.get("HIAST_LIB_WGRAD3", "0") != "1"


def conv_wgrad_small_supported(Cin, Cout, k, stride):
    """shapes hiast_conv_wgrad_small_nhwc takes (K9h: the layers below 256 channels; a strided 1x1 runs on the subsampled
    input); shapes hiast_conv_wgrad_nhwc takes as well stay there"""
    import os

    def chan(c):
        return c == 64 or (c >= 128 and c % 128 == 0)
    if os.environ.get("HIAST_LIB_WGRAD_SMALL", "0") == "1" or conv_wgrad_supported(Cin, Cout, k, stride):
        return False
    return chan(Cin) and chan(Cout) and ((k == 1 and stride in (1, 2)) or k == 3)


def conv_wgrad_small_nhwc(dy, x, k, stride, dil):
    """dy [B,Ho,Wo,Cout], x [B,H,W,Cin] bf16 / fp16 channels-last rows -> dW fp32 [Cout,Cin,k,k] (K9h)"""
    _req16(dy, 4, "dy")
    _req(x, dy.dtype, 4, "x")
    if stride != 1 and k == 1:
        x = x[:, ::stride, ::stride, :].contiguous()         # a strided 1x1 sees every stride-th pixel only
        stride = 1
    B, H, W, Cin = x.shape
    Bo, Ho, Wo, Cout = dy.shape
    assert (Bo, Ho, Wo) == (B, (H - 1) // stride + 1, (W - 1) // stride + 1)
    taps = k * k
    lib = _lib.load()
    n = lib.hiast_conv_wgrad_small_workspace_bytes(B, Ho, Wo, Cin, Cout, taps)
    if n == 0:
        raise _lib.HiastLibraryError("hiast_conv_wgrad_small_nhwc: unsupported shape Cin=%d Cout=%d taps=%d" % (Cin, Cout, taps))
    # one growing scratch buffer per (device, stream): launches of ONE stream reuse it in stream order; the weight gradients
    # of a backward pass run on two streams (main for the layers whose data gradient is the library's, the side stream for
    # the rest) and two launches that shared the partials would overwrite each other's
    key = ("small", x.device, torch.cuda.current_stream(x.device).cuda_stream)
    ws = _wgrad_ws.get(key)
    if ws is None or ws.numel() * 4 < n:
        ws = torch.empty((n + 3) // 4, dtype=torch.float32, device=x.device)
        _wgrad_ws[key] = ws
    dw = torch.empty((Cout, Cin, k, k), dtype=torch.float32, device=x.device)
    check(lib.hiast_conv_wgrad_small_nhwc(_ptr(dy), _ptr(x), _ptr(dw), B, H, W, Cin, Cout, taps, int(stride), int(dil), fmt_of(dy),
                                          _ptr(ws), ws.numel() * 4, _stream()), "hiast_conv_wgrad_small_nhwc")
    return dw


def conv_wgrad_nhwc(dy, x, k, stride, dil):
    """dy [B,Ho,Wo,Cout], x [B,H,W,Cin] bf16 / fp16 channels-last rows (same type) -> dW fp32 [Cout,Cin,k,k]"""
    _req16(dy, 4, "dy")
    _req(x, dy.dtype, 4, "x")
    B, H, W, Cin = x.shape
    Bo, Ho, Wo, Cout = dy.shape
    taps = k * k
    assert Bo == B and (Ho, Wo) == ((H, W) if k == 1 else ((H - 1) // stride + 1, (W - 1) // stride + 1))
    lib = _lib.load()
    n = lib.hiast_conv_wgrad_workspace_bytes(B, Ho, Wo, Cin, Cout, taps)
    if n == 0:
        raise _lib.HiastLibraryError("hiast_conv_wgrad_nhwc: unsupported shape Cin=%d Cout=%d taps=%d" % (Cin, Cout, taps))
    key = ("big", x.device, torch.cuda.current_stream(x.device).cuda_stream)     # per stream: see conv_wgrad_small_nhwc
    ws = _wgrad_ws.get(key)
    if ws is None or ws.numel() * 4 < n:
        ws = torch.empty((n + 3) // 4, dtype=torch.float32, device=x.device)
        _wgrad_ws[key] = ws
    dw = torch.empty((Cout, Cin, k, k), dtype=torch.float32, device=x.device)
    check(lib.hiast_conv_wgrad_nhwc(_ptr(dy), _ptr(x), _ptr(dw), B, H, W, Cin, Cout, taps, int(stride), int(dil), fmt_of(dy),
                                    _ptr(ws), ws.numel() * 4, _stream()), "hiast_conv_wgrad_nhwc")
    return dw


def conv_wgrad_group(jobs):
    """the weight gradients of up to 4 convolutions in ONE launch + ONE reduction (hiast_conv_wgrad_group_nhwc): jobs =
    [(dy [B,Ho,Wo,Cout], x [B,H,W,Cin], k, stride, dil), ...], 16-bit channels-last rows of one type, every shape one that
    conv_wgrad_supported() takes -> [dW fp32 [Cout,Cin,k,k], ...].  The jobs share the pixel-range split: the chip is
    filled once for all of them (one fp32 partial tile per CU) instead of once per convolution."""
    import ctypes
    lib = _lib.load()
    n = len(jobs)
    arr = (_lib.WgradJob * n)()
    outs = []
    fmt = None
    for i, (dy, x, k, stride, dil) in enumerate(jobs):
        _req16(dy, 4, "dy")
        _req(x, dy.dtype, 4, "x")
        B, H, W, Cin = x.shape
        Bo, Ho, Wo, Cout = dy.shape
        assert Bo == B and (Ho, Wo) == ((H, W) if k == 1 else ((H - 1) // stride + 1, (W - 1) // stride + 1))
        assert fmt is None or fmt == fmt_of(dy), "one operand type per grouped launch"
        fmt = fmt_of(dy)
        dw = torch.empty((Cout, Cin, k, k), dtype=torch.float32, device=x.device)
        outs.append(dw)
        arr[i] = _lib.WgradJob(_ptr(dy), _ptr(x), _ptr(dw), B, H, W, Cin, Cout, k * k, int(stride), int(dil))
    need = lib.hiast_conv_wgrad_group_workspace_bytes(ctypes.addressof(arr), n)
    if need == 0:
        raise _lib.HiastLibraryError("hiast_conv_wgrad_group_nhwc: unsupported shapes %s" %
                                     [(tuple(j[0].shape), tuple(j[1].shape), j[2]) for j in jobs])
    dev = jobs[0][1].device
    key = ("group", dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _wgrad_ws.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=dev)
        _wgrad_ws[key] = ws
    check(lib.hiast_conv_wgrad_group_nhwc(ctypes.addressof(arr), n, fmt, _ptr(ws), ws.numel() * 4, _stream()),
          "hiast_conv_wgrad_group_nhwc")
    return outs
