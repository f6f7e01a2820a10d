"""Parity oracle for the HIAST hot path — TEST INFRASTRUCTURE ONLY.

CPU restatements (plain C in hiast_oracle.c, numpy / torch-CPU fp32 here) of the reference's
algorithm, each function citing the reference file:line it follows, pinned against fixtures
produced by running the reference itself (tests/golden/).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this package; hiast_amd/ never does.
"""
