"""CPU-side checks of the drop-in boundary: libhiast_hip.so loads and exports every symbol that
include/hiast_hip.h declares (no compute: there is no GPU here)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "hiast_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hiast_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from hiast_amd import _lib
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), s
    assert sorted(_lib.SIGNATURES) == syms
    assert lib.hiast_version() == 5
    assert b"workspace" in lib.hiast_error_string(-3)


def declared_arity():
    """name -> number of parameters of its declaration in include/hiast_hip.h"""
    txt = open(os.path.join(ROOT, "include", "hiast_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(hiast_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_ctypes_signatures_match_the_header_arity():
    """a ctypes argtypes list that is one argument short or long does not fail at load time — it shifts every later
    argument of the call (a stream handle read as a size, ...): every binding must have the declaration's arity"""
    from hiast_amd import _lib
    arity = declared_arity()
    assert sorted(arity) == sorted(_lib.SIGNATURES)
    for name, (_res, args) in _lib.SIGNATURES.items():
        assert len(args) == arity[name], (name, len(args), arity[name])


def test_host_side_sizes():
    from hiast_amd import _lib
    lib = _lib.load()
    assert lib.hiast_aspp_wpack_bytes(2048, 19) == (33 * 2048 * 32 + 32) * 4
    assert lib.hiast_st_loss_workspace_bytes(8, 19, 64, 128, 512, 1024) > 0
    assert lib.hiast_st_loss_workspace_bytes(1, 19, 2, 2, 1024, 2048) == 0   # > 253x upsampling: refused
