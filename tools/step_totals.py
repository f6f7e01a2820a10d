"""profiles/r06_step_totals.json — the whole-step figures bench.py quotes as `step_frac_hbm`, `serial_kernel_ms`,
`dispatches_per_step` (they need profiler passes of their own and cannot be collected inside a bench run):
    python tools/step_totals.py gpurun_out/pmcb_<tag> gpurun_out/prof_<tag>_serial profiles/r06_step_totals.json
  pmcb dir: tools/pmc_bench.sh (two rocprofv3 --pmc passes over the whole bench process: FETCH_SIZE.csv, WRITE_SIZE.csv + the
            bench's JSON line of the FETCH_SIZE pass, which says how many steps the process ran)
  prof dir: tools/prof_bench.sh <tag>_serial (rocprofv3 --kernel-trace: steady_kernels.csv)
fabric bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per dispatch (MI355X_MICROARCH.md: FETCH_SIZE counts 128-B requests as 64 B on
gfx950; Infinity-Cache hits included: an upper bound on HBM bytes), summed over the dispatches of the timed steps."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def total_kb(path):
    """counter sum over the dispatches of the TIMED region: between the two hiast::confusion_kernel markers bench.py launches
    around it (tools/trace_summary.py uses the same markers) — the one-time state preparation of the bench (fp32 torch modules on
    the library's naive convolutions: 10 % of the process's bytes) stays out"""
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "confusion_kernel" in r["Kernel_Name"]]
    assert len(marks) >= 2, "markers not found"
    return sum(float(r["Counter_Value"]) for r in rows[marks[0] + 1:marks[1]])


def main():
    pmcb, prof, dst = sys.argv[1], sys.argv[2], sys.argv[3]
    line = json.loads(open(os.path.join(pmcb, "FETCH_SIZE.json")).read().strip().splitlines()[-1])
    steps = line["steps"]                       # the timed steps between the markers
    fetch, write = total_kb(os.path.join(pmcb, "FETCH_SIZE.csv")), total_kb(os.path.join(pmcb, "WRITE_SIZE.csv"))
    gb = (2.0 * fetch + write) * 1024.0 / steps / 1e9
    busy = disp = None
    for r in csv.reader(open(os.path.join(prof, "steady_kernels.csv"))):
        if r and r[0] == "TOTAL_gpu_busy":
            busy = float(r[4])
        if r and r[0] == "DISPATCHES_per_step":
            disp = float(r[2])
    from hiast_amd import _lib
    out = {"kernel_sources_sha16": _lib.kernel_sources_sha16(),
           "source": "tools/pmc_bench.sh (the %d timed steps between bench.py's markers) + tools/prof_bench.sh serial; tools/step_totals.py" % steps,
           "hbm_gb_per_step": gb, "read_gb_per_step": 2.0 * fetch * 1024.0 / steps / 1e9,
           "written_gb_per_step": write * 1024.0 / steps / 1e9, "serial_kernel_ms": busy, "dispatches_per_step": disp,
           "correction": "(2*FETCH_SIZE + WRITE_SIZE)*1024, gfx950"}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
