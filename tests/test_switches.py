"""hiast_amd/switches.py: the path-selecting switches are read once at import (a per-call lookup could differ between the ranks of
a job, or between a forward and its backward: mismatched collective order under SyncBN) and flipped by tests through the dict."""
import importlib
import os


def test_switches_are_read_once_at_import(monkeypatch):
    from hiast_amd import switches as SW
    assert set(SW.SWITCHES) == {"HIAST_NO_BN_MASK", "HIAST_NO_BN_BWD_FUSION", "HIAST_LIB_WGRAD", "HIAST_NO_WGROUP",
                                "HIAST_NO_XSUM", "HIAST_NO_IDT_HANDOFF", "HIAST_LIB_STEM"}
    before = dict(SW.SWITCHES)
    monkeypatch.setenv("HIAST_NO_XSUM", "0" if before["HIAST_NO_XSUM"] else "1")
    assert SW.on("HIAST_NO_XSUM") == before["HIAST_NO_XSUM"]          # the environment is not consulted again
    monkeypatch.setitem(SW.SWITCHES, "HIAST_NO_XSUM", not before["HIAST_NO_XSUM"])
    assert SW.on("HIAST_NO_XSUM") != before["HIAST_NO_XSUM"]


def test_switches_follow_the_environment_of_the_import(monkeypatch):
    monkeypatch.setenv("HIAST_NO_WGROUP", "1")
    monkeypatch.setenv("HIAST_LIB_STEM", "0")
    import hiast_amd.switches as SW
    fresh = importlib.reload(SW)
    try:
        assert fresh.on("HIAST_NO_WGROUP") is True and fresh.on("HIAST_LIB_STEM") is False
    finally:
        monkeypatch.delenv("HIAST_NO_WGROUP")
        monkeypatch.delenv("HIAST_LIB_STEM")
        importlib.reload(SW)
    assert os.environ.get("HIAST_NO_WGROUP") is None
