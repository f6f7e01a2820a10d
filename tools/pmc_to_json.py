"""tools/pmc_igemm.sh summary.txt -> profiles/<tag>_pmc_igemm.json (the file bench.py reads roofline.traffic from).
    python3 tools/pmc_to_json.py gpurun_out/pmc_igemm/summary.txt profiles/r02_pmc_igemm.json"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_igemm import SHAPES      # noqa: E402  (imports torch; host only)


def main():
    src, dst = sys.argv[1], sys.argv[2]
    cur, vals, alg = None, {}, {}
    for line in open(src):
        m = re.match(r"== (\S+) \|", line)
        if m:
            cur = m.group(1)
            vals.setdefault(cur, {})
            continue
        m = re.match(r"\s+(\w+)\s+n=\s*\d+ mean=([0-9.e+-]+)", line)
        if m and cur:
            vals[cur][m.group(1)] = float(m.group(2))
            continue
        m = re.match(r"shape (\S+) algorithmic_bytes (\d+)", line)
        if m:
            alg[m.group(1)] = int(m.group(2))
    kernels = []
    for name, v in vals.items():
        B, H, W, Cin, Cout, taps, dil, PL, has_res = SHAPES[name]
        if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        ent = {"name": name, "key": [B, H, W, Cin, Cout, taps, PL, dil], "has_res": has_res,
               "FETCH_SIZE_KB": v["FETCH_SIZE"], "WRITE_SIZE_KB": v["WRITE_SIZE"],
               "hbm_bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024,
               "hbm_bytes_uncorrected": (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024,
               "algorithmic_bytes": alg.get(name)}
        if "TCC_HIT_sum" in v:
            ent["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
        for c in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
            if c in v:
                ent[c] = v[c]
        kernels.append(ent)
    from hiast_amd import _lib
    out = {"kernel_sources_sha16": _lib.kernel_sources_sha16(),      # bench.py flags `traffic` as stale on another build
           "source": "%s (tools/pmc_igemm.sh on the kernels of the tree at collection time)" % os.path.basename(src),
           "command": "tools/pmc_igemm.sh: rocprofv3 --pmc <one counter group per pass> --output-format csv -- python3 "
                      "tools/pmc_igemm.py <shape>; means over 6 launches per shape",
           "correction": "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE counts 128-B fabric "
                         "requests as 64 B (MI355X_MICROARCH.md, HBM section); Infinity-Cache hits are included, so this is "
                         "an upper bound on HBM bytes",
           "kernels": kernels}
    json.dump(out, open(dst, "w"), indent=1)
    print("wrote", dst, [k["name"] for k in kernels])


if __name__ == "__main__":
    main()
