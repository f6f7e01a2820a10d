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
    assert lib.hiast_version() == 6
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


def test_thread_local_hints_and_cu_reserve_are_host_state():
    """ABI 6: the co-scheduling hint and the tile-form override are state of the CALLING thread (no environment variable is
    written at run time: setenv raced with getenv in other threads), the CU reserve is process-wide; none of them needs a GPU"""
    import threading
    from hiast_amd import _lib
    lib = _lib.load()
    assert lib.hiast_igemm_set_cosched(1) == -1          # default: unset
    assert lib.hiast_igemm_set_half(0) == -1
    seen = []
    t = threading.Thread(target=lambda: seen.append((lib.hiast_igemm_set_cosched(-1), lib.hiast_igemm_set_half(-1))))
    t.start(); t.join()
    assert seen == [(-1, -1)]                            # another thread does not see this thread's hint
    M = 4 * 64 * 128                                     # 1024 -> 256 on a 4-image batch: 128 tiles of 256 rows (half the chip)
    lib.hiast_igemm_set_half(-1)
    assert lib.hiast_igemm_stats_rows(M, 1024, 256, 1, 3) == M // 256      # co-scheduled: the 256-row form
    assert lib.hiast_igemm_set_cosched(-1) == 1
    assert lib.hiast_igemm_stats_rows(M, 1024, 256, 1, 3) == M // 128      # alone: 128 x 128 tiles, two blocks per CU
    assert lib.hiast_igemm_set_half(0) == -1
    assert lib.hiast_igemm_stats_rows(M, 1024, 256, 1, 3) == M // 256
    assert lib.hiast_igemm_dgrad_bn_stats_rows(M, 1024, 256, 1) == M // 256
    lib.hiast_igemm_set_half(-1)
    # the CU reserve: rounded up to whole rounds of the 8 XCDs, bounded by half of the device, returns the previous value
    assert lib.hiast_get_reserve_cus() == 0
    r0 = lib.hiast_igemm_stats_rows(8 * 64 * 128, 256, 1024, 1, 3)          # xconv: 256 CUs / column groups
    assert r0 in (64, 128)
    assert lib.hiast_set_reserve_cus(5) == 0 and lib.hiast_get_reserve_cus() == 8
    # (persistent launches now plan for 248 CUs: the xconv row count — one row per block row of a 256-column group — follows)
    assert lib.hiast_igemm_stats_rows(8 * 64 * 128, 256, 1024, 1, 3) == (248 // (256 // r0)) // 8 * 8
    assert lib.hiast_set_reserve_cus(0) == 8
    assert lib.hiast_igemm_stats_rows(8 * 64 * 128, 256, 1024, 1, 3) == r0
    assert lib.hiast_set_reserve_cus(-1) == -1 and lib.hiast_set_reserve_cus(200) == -2


def test_host_side_sizes():
    from hiast_amd import _lib
    lib = _lib.load()
    assert lib.hiast_aspp_wpack_bytes(2048, 19) == (33 * 2048 * 32 + 32) * 4
    assert lib.hiast_st_loss_workspace_bytes(8, 19, 64, 128, 512, 1024) > 0
    assert lib.hiast_st_loss_workspace_bytes(1, 19, 2, 2, 1024, 2048) == 0   # > 253x upsampling: refused
