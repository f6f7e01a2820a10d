"""ctypes binding of libhiast_hip.so (the C ABI of include/hiast_hip.h).

There is NO fallback: if the library is missing or an entry point returns non-zero, the call
raises.  `import torch` happens first so the library binds to the same libamdhip64 that
PyTorch-ROCm has loaded (device pointers and streams are shared).
"""
import ctypes
import os

import torch  # noqa: F401  (loads libamdhip64 before our library is opened)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HIAST_LIB") or os.path.join(_HERE, "csrc", "libhiast_hip.so")   # HIAST_LIB: A/B of two builds

c_int, c_i64, c_f32, c_sz, c_vp = (ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t,
                                   ctypes.c_void_p)

# name -> (restype, argtypes); mirrors include/hiast_hip.h one to one
SIGNATURES = {
    "hiast_version": (c_int, []),
    "hiast_error_string": (ctypes.c_char_p, [c_int]),
    "hiast_upsample_bilinear_ac_fwd": (c_int, [c_vp, c_vp] + [c_int] * 6 + [c_vp]),
    "hiast_upsample_bilinear_ac_bwd": (c_int, [c_vp, c_vp] + [c_int] * 6 + [c_vp]),
    "hiast_plabel_pass1_workspace_bytes": (c_sz, [c_int]),
    "hiast_plabel_pass1": (c_int, [c_vp] + [c_int] * 6 + [c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "hiast_plabel_pass2": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "hiast_plabel_strided_hist_workspace_bytes": (c_sz, [c_i64, c_int]),
    "hiast_plabel_strided_hist": (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "hiast_tta_fused": (c_int, [c_vp] * 6 + [c_int] * 5 + [c_vp, c_vp, c_vp]),
    "hiast_dinput_fwd": (c_int, [c_vp, c_int, c_vp] + [c_int] * 6 + [c_vp]),
    "hiast_dinput_bwd": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp] + [c_int] * 6 + [c_vp]),
    "hiast_st_loss_workspace_bytes": (c_sz, [c_int] * 6),
    "hiast_st_loss_fwd": (c_int, [c_vp, c_vp, c_vp] + [c_int] * 8 + [c_vp, c_vp, c_sz, c_vp]),
    "hiast_st_loss_bwd": (c_int, [c_vp, c_vp, c_vp] + [c_int] * 8 + [c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "hiast_aspp_wpack_bytes": (c_sz, [c_int, c_int]),
    "hiast_aspp_workspace_bytes": (c_sz, [c_int] * 5),
    "hiast_aspp_pack_weights": (c_int, [c_vp] * 8 + [c_int, c_int, c_vp, c_vp]),
    "hiast_aspp_fwd": (c_int, [c_vp, c_vp, c_vp] + [c_int] * 5 + [c_vp, c_vp, c_sz, c_vp]),
    "hiast_aspp_bwd_data": (c_int, [c_vp, c_vp, c_vp] + [c_int] * 5 + [c_vp, c_vp]),
    "hiast_aspp_bwd_weight": (c_int, [c_vp] * 7 + [c_int] * 5 + [c_vp, c_vp, c_sz, c_vp]),
    "hiast_aspp2_np": (c_int, [c_int]),
    "hiast_aspp2_workspace_bytes": (c_sz, [c_int] * 6),
    "hiast_aspp2_pack_weights": (c_int, [c_vp] * 8 + [c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "hiast_aspp2_fwd": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp] + [c_int] * 5 + [c_vp, c_vp, c_sz, c_vp]),
    "hiast_aspp2_bwd": (c_int, [c_vp] * 9 + [c_int] * 5 + [c_vp, c_int, c_vp, c_sz, c_vp]),
    "hiast_bn_workspace_bytes": (c_sz, [c_int, c_int]),
    "hiast_bn_stats": (c_int, [c_vp, c_int, c_int, c_i64, c_int, c_vp, c_vp]),
    "hiast_bn_act_apply": (c_int, [c_vp] * 8 + [c_int, ctypes.c_double, c_f32, c_f32, c_int, c_vp, c_vp,
                                   c_int, c_int, c_i64, c_int, c_vp]),
    "hiast_bn_act_bwd_stats": (c_int, [c_vp] * 5 + [c_int, c_int, c_int, c_i64, c_int, c_vp, c_vp]),
    "hiast_bn_act_bwd_apply": (c_int, [c_vp] * 7 + [c_int, ctypes.c_double, c_int, c_vp, c_vp, c_vp, c_vp,
                                       c_int, c_int, c_i64, c_int, c_vp]),
    "hiast_bn_act_nhwc_infer": (c_int, [c_vp] * 6 + [c_f32, c_int, c_i64, c_int, c_int, c_vp]),
    "hiast_igemm_bn_act": (c_int, [c_vp] * 6 + [c_f32, c_vp, c_int, c_vp] + [c_int] * 10 + [c_vp, c_int, c_vp, c_int, c_vp]),
    "hiast_igemm_set_cosched": (c_int, [c_int]),
    "hiast_igemm_set_half": (c_int, [c_int]),
    "hiast_device_cus": (c_int, []),
    "hiast_set_reserve_cus": (c_int, [c_int]),
    "hiast_get_reserve_cus": (c_int, []),
    "hiast_stream_create_reserved": (c_int, [c_vp, c_int]),
    "hiast_stream_destroy": (c_int, [c_vp]),
    "hiast_igemm_stats_rows": (c_int, [c_i64, c_int, c_int, c_int, c_int]),
    "hiast_igemm_dgrad_bn_stats": (c_int, [c_vp, c_vp, c_vp] + [c_int] * 7 + [c_vp] * 6 + [c_int, c_int, c_vp]),
    "hiast_xconv_dgrad_gated_bn_stats_rows": (c_int, [c_i64, c_int, c_int]),
    "hiast_xconv_dgrad_gated_bn_stats": (c_int, [c_vp] * 10 + [c_i64, c_int, c_int, c_int, c_vp]),
    "hiast_igemm_dgrad_s2": (c_int, [c_vp, c_vp, c_vp] + [c_int] * 6 + [c_vp]),
    "hiast_igemm_dgrad_bn_stats_rows": (c_int, [c_i64, c_int, c_int, c_int]),
    "hiast_bn_nhwc_stats_from_partial": (c_int, [c_vp, c_int, c_int, c_vp, c_vp]),
    "hiast_conv_wgrad_workspace_bytes": (c_sz, [c_int] * 6),
    "hiast_conv_wgrad_nhwc": (c_int, [c_vp, c_vp, c_vp] + [c_int] * 9 + [c_vp, c_sz, c_vp]),
    "hiast_conv_wgrad_group_workspace_bytes": (c_sz, [c_vp, c_int]),
    "hiast_conv_wgrad_group_nhwc": (c_int, [c_vp, c_int, c_int, c_vp, c_sz, c_vp]),
    "hiast_conv_wgrad_small_workspace_bytes": (c_sz, [c_int] * 6),
    "hiast_conv_wgrad_small_nhwc": (c_int, [c_vp, c_vp, c_vp] + [c_int] * 9 + [c_vp, c_sz, c_vp]),
    "hiast_pack_conv_weight": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "hiast_bn_nhwc_workspace_bytes": (c_sz, [c_int]),
    "hiast_bn_nhwc_stats": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp, c_sz, c_int, c_vp]),
    "hiast_bn_nhwc_apply": (c_int, [c_vp] * 8 + [ctypes.c_double, c_f32, c_f32, c_int, c_vp, c_vp, c_i64, c_int, c_vp,
                                    c_int, c_vp]),
    "hiast_bn_nhwc_apply_partial": (c_int, [c_vp] * 8 + [c_int, ctypes.c_double, c_f32, c_f32, c_int, c_vp, c_vp, c_i64,
                                            c_int, c_vp, c_int, c_vp]),
    "hiast_bn_nhwc_bwd_stats": (c_int, [c_vp] * 7 + [c_int, c_i64, c_int, c_vp, c_vp, c_sz, c_int, c_vp]),
    "hiast_bn_nhwc_bwd_apply": (c_int, [c_vp] * 8 + [ctypes.c_double, c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int,
                                        c_vp]),
    "hiast_pack_conv_weight_multi": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_vp]),
    "hiast_split_planes": (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp]),
    "hiast_stem_tail": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "hiast_stem_eval": (c_int, [c_vp] * 6 + [c_f32, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "hiast_stem_train_blocks": (c_int, [c_int] * 3),
    "hiast_stem_train_fwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "hiast_stem_wgrad_workspace_bytes": (c_sz, [c_int] * 3),
    "hiast_stem_wgrad": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_sz, c_vp]),
    "hiast_ema_update": (c_int, [c_vp, c_vp, c_vp, c_int, c_f32, c_f32, c_vp]),
    "hiast_normalize_u8": (c_int, [c_vp, c_vp, c_int, c_i64, c_vp, c_vp, c_vp]),
    "hiast_maxpool3x3s2_nhwc_fwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "hiast_maxpool3x3s2_nhwc_bwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "hiast_multi_copy": (c_int, [c_vp, c_int, c_vp]),
    "hiast_adam_prepare": (c_int, [c_vp, c_vp, c_vp, c_vp]),
    "hiast_adam_step": (c_int, [c_vp, c_vp, c_vp, c_int, ctypes.c_double, ctypes.c_double, c_f32, c_f32, c_vp, c_vp]),
    "hiast_confusion_hist": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp]),
}



class WgradJob(ctypes.Structure):
    """hiast_wgrad_job of include/hiast_hip.h"""
    _fields_ = [("dy", c_vp), ("x", c_vp), ("dw", c_vp)] + [(n, ctypes.c_int32) for n in
                                                             ("B", "H", "W", "Cin", "Cout", "taps", "stride", "dil")]


NBINS = 15361
PROB_FX_SHIFT = 30
IGNORE = 255

_lib = None


class HiastLibraryError(RuntimeError):
    pass


def load():
    """Open libhiast_hip.so and type every entry point; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HiastLibraryError(
            "libhiast_hip.so is not built (%s). Build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C hiast_amd/csrc`; there is no fallback path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HiastLibraryError("libhiast_hip.so does not export %s" % name) from e
        fn.restype = res
        fn.argtypes = args
    if lib.hiast_version() != 6:
        raise HiastLibraryError("libhiast_hip.so ABI version mismatch")
    _lib = lib
    return lib


def kernel_sources_sha16():
    """fingerprint of the kernel sources the library is built from (hiast_amd/csrc/*.hip, *.h + the ABI header): what a
    counter collection (tools/pmc_to_json.py) records, and what bench.py compares before it quotes bytes measured on
    another build (`roofline.traffic` read from a file of an earlier tree)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "hiast_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def check(code, what):
    if code != 0:
        msg = load().hiast_error_string(int(code)).decode()
        raise HiastLibraryError("%s failed: %s (code %d)" % (what, msg, code))
