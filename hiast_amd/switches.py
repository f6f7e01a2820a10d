"""Path-selecting environment switches of the training step, read ONCE at import.

Each of them changes which autograd path (and, under SyncBN, which sequence of collectives) a forward / backward takes.  Looked
up per call they could differ between the ranks of a job, or between a forward and its backward — mismatched collective order on
the statistics communicator is a hang.  Tests flip them with `monkeypatch.setitem(switches.SWITCHES, name, value)`."""
import os

_NAMES = ("HIAST_NO_BN_MASK", "HIAST_NO_BN_BWD_FUSION", "HIAST_LIB_WGRAD", "HIAST_NO_WGROUP", "HIAST_NO_XSUM",
          "HIAST_NO_IDT_HANDOFF", "HIAST_LIB_STEM")
SWITCHES = {n: os.environ.get(n, "0") == "1" for n in _NAMES}


def on(name):
    return SWITCHES[name]
