"""ctypes/numpy front-end of oracle/hiast_oracle.c.

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by anything under hiast_amd/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_u64p = ctypes.POINTER(ctypes.c_uint64)


def build(force=False):
    """Compile oracle/_build/liboracle.so with gcc (see oracle/Makefile)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_HERE, "hiast_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_expf.restype = ctypes.c_float
        _lib.orc_expf.argtypes = [ctypes.c_float]
        _lib.orc_f16_bits.restype = ctypes.c_uint16
        _lib.orc_f16_bits.argtypes = [ctypes.c_float]
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(ty)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def expf(x):
    L = lib()
    x = np.asarray(x, np.float32)
    return np.array([L.orc_expf(float(v)) for v in x.ravel()], np.float32).reshape(x.shape)


def upsample_bilinear_ac(x, H, W):
    x = _c(x, np.float32)
    B, C, h, w = x.shape
    H, W = int(H), int(W)
    out = np.empty((B, C, H, W), np.float32)
    lib().orc_upsample_bilinear_ac(_p(x, _f32p), _p(out, _f32p), B * C, h, w, int(H), int(W))
    return out


def upsample_bilinear_ac_bwd(gout, h, w):
    gout = _c(gout, np.float32)
    B, C, H, W = gout.shape
    h, w = int(h), int(w)
    gin = np.empty((B, C, h, w), np.float32)
    lib().orc_upsample_bilinear_ac_bwd(_p(gout, _f32p), _p(gin, _f32p), B * C, h, w, H, W)
    return gin


def plabel_stage_a(logits_lr, H, W):
    """low-res logits [B,C,h,w] -> (maxprob f32 [B,H,W], argmax u8 [B,H,W])"""
    z = _c(logits_lr, np.float32)
    B, C, h, w = z.shape
    H, W = int(H), int(W)
    mp = np.empty((B, H, W), np.float32)
    am = np.empty((B, H, W), np.uint8)
    lib().orc_plabel_stage_a(_p(z, _f32p), B, C, h, w, H, W, _p(mp, _f32p), _p(am, _u8p))
    return mp, am


def plabel_hist(maxprob, argmax, C, nbins=15361):
    mp = _c(maxprob, np.float32)
    am = _c(argmax, np.uint8)
    C = int(C)
    hist = np.zeros((C, nbins), np.uint32)
    lib().orc_plabel_hist(_p(mp, _f32p), _p(am, _u8p), ctypes.c_int64(mp.size), int(C), int(nbins),
                          _p(hist, _u32p))
    return hist


def plabel_select(maxprob, argmax, thr, C):
    """thr: float64 [C] or None -> (plbl u8 [B,H,W], count i64 [B,C], sumprob_fx u64 [C])"""
    mp = _c(maxprob, np.float32)
    am = _c(argmax, np.uint8)
    C = int(C)
    B = mp.shape[0]
    HW = mp.size // B
    plbl = np.empty(mp.shape, np.uint8)
    count = np.zeros((B, C), np.int64)
    sfx = np.zeros((C,), np.uint64)
    tp = None
    if thr is not None:
        thr = _c(thr, np.float64)
        tp = _p(thr, _f64p)
    lib().orc_plabel_select(_p(mp, _f32p), _p(am, _u8p), tp, B, C, ctypes.c_int64(HW),
                            _p(plbl, _u8p), _p(count, _i64p), _p(sfx, _u64p))
    return plbl, count, sfx


def aspp_fwd(x, weights, biases, dil):
    x = _c(x, np.float32)
    B, Cin, h, w = x.shape
    ws = [_c(wi, np.float32) for wi in weights]
    bs = [_c(bi, np.float32) for bi in biases]
    Cout = ws[0].shape[0]
    y = np.empty((B, Cout, h, w), np.float32)
    WP = (_f32p * 4)(*[_p(wi, _f32p) for wi in ws])
    BP = (_f32p * 4)(*[_p(bi, _f32p) for bi in bs])
    d = (ctypes.c_int * 4)(*[int(v) for v in dil])
    lib().orc_aspp_fwd(_p(x, _f32p), WP, BP, _p(y, _f32p), B, Cin, h, w, Cout, d)
    return y


def confusion_hist(pred, target, K):
    p = _c(pred, np.int64).ravel()
    t = _c(target, np.int64).ravel()
    K = int(K)
    inter = np.zeros(K, np.int64)
    ap = np.zeros(K, np.int64)
    at = np.zeros(K, np.int64)
    lib().orc_confusion_hist(_p(p, _i64p), _p(t, _i64p), ctypes.c_int64(p.size), K,
                             _p(inter, _i64p), _p(ap, _i64p), _p(at, _i64p))
    return inter, ap, at


def ema_update(ema, p, gamma):
    """in-place on a float32 numpy array `ema`; gamma is the Python double of the config"""
    assert ema.dtype == np.float32 and ema.flags.c_contiguous
    p = _c(p, np.float32)
    lib().orc_ema_update(_p(ema, _f32p), _p(p, _f32p), ctypes.c_int64(ema.size),
                         ctypes.c_float(np.float32(gamma)), ctypes.c_float(np.float32(1 - gamma)))
    return ema


def tta(zs, zfs, sizes, H, W, want_probs=True):
    """multi-scale + flip TTA from low-res head outputs (orc_tta): zs / zfs lists of [B,C,hs,ws] (zfs entries may be
    None = no flip), sizes list of (Hs, Ws) -> (probsum f32 [B,C,H,W] or None, label u8 [B,H,W])"""
    n = len(zs)
    zs = [_c(z, np.float32) for z in zs]
    B, C = zs[0].shape[:2]
    flip = zfs is not None and any(z is not None for z in zfs)
    zfs = [_c(z, np.float32) if z is not None else None for z in zfs] if flip else None
    ZP = (_f32p * n)(*[_p(z, _f32p) for z in zs])
    ZF = (_f32p * n)(*[(_p(z, _f32p) if z is not None else None) for z in zfs]) if flip else None
    iarr = lambda v: (ctypes.c_int * n)(*[int(x) for x in v])
    H, W = int(H), int(W)
    probs = np.empty((B, C, H, W), np.float32) if want_probs else None
    label = np.empty((B, H, W), np.uint8)
    lib().orc_tta(ZP, ZF, iarr([z.shape[2] for z in zs]), iarr([z.shape[3] for z in zs]), iarr([s[0] for s in sizes]),
                  iarr([s[1] for s in sizes]), n, B, C, H, W, _p(probs, _f32p) if want_probs else None, _p(label, _u8p))
    return probs, label
