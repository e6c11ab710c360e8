"""Summarise rocprofv3 --pmc output (…_counter_collection.csv) per kernel family: launches, counter sums, mean per
launch, duration.  FETCH_SIZE is doubled (gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md, HBM section);
FETCH_SIZE / WRITE_SIZE are reported in bytes (the CSV holds KiB).  Usage:
    python tools/pmc_summary.py gpurun_out/r2/pmc_FETCH_SIZE [more dirs...] > profiles/r2_pmc_summary.json
Kernel names are reduced to `family<template args>` so the split GEMM tile variants stay distinguishable."""
import csv
import json
import re
import sys
from collections import defaultdict
from pathlib import Path


def family(name: str) -> str:
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void\s+", "", n)
    n = n.split("(")[0].strip()
    return n[:90]


def summarise(path: Path):
    out = defaultdict(lambda: defaultdict(lambda: [0, 0.0, 0]))      # family -> counter -> [launch rows, sum, ns]
    for f in path.rglob("*counter_collection.csv"):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                e = out[family(r["Kernel_Name"])][r["Counter_Name"]]
                e[0] += 1
                e[1] += float(r["Counter_Value"])
                e[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    res = {}
    for fam, ctrs in out.items():
        d = {}
        for c, (n, v, ns) in ctrs.items():
            scale = 1.0
            if c == "FETCH_SIZE":
                scale = 2 * 1024.0
            elif c == "WRITE_SIZE":
                scale = 1024.0
            d[c] = {"launches": n, "sum": v * scale, "mean_per_launch": v * scale / max(n, 1), "total_ms": ns / 1e6}
        res[fam] = d
    return res


def main():
    merged = {}
    for a in sys.argv[1:]:
        for fam, d in summarise(Path(a)).items():
            merged.setdefault(fam, {}).update(d)
    top = sorted(merged.items(), key=lambda kv: -max(v["total_ms"] for v in kv[1].values()))[:25]
    derived = {}
    for fam, d in top:
        x = {}
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            ms = d["FETCH_SIZE"]["total_ms"]
            x["hbm_TBps"] = (d["FETCH_SIZE"]["sum"] + d["WRITE_SIZE"]["sum"]) / (ms / 1e3) / 1e12 if ms else None
        if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
            h, m = d["TCC_HIT_sum"]["sum"], d["TCC_MISS_sum"]["sum"]
            x["l2_hit_rate"] = h / max(h + m, 1)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d:
            x["mfma_busy_over_sq_busy"] = d["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / max(d["SQ_BUSY_CYCLES"]["sum"], 1)
        if "SQ_WAIT_ANY" in d and "SQ_WAVE_CYCLES" in d:
            x["wait_any_over_wave_cycles"] = d["SQ_WAIT_ANY"]["sum"] / max(d["SQ_WAVE_CYCLES"]["sum"], 1)
        if "SQ_WAIT_INST_LDS" in d and "SQ_WAVE_CYCLES" in d:
            x["wait_lds_over_wave_cycles"] = d["SQ_WAIT_INST_LDS"]["sum"] / max(d["SQ_WAVE_CYCLES"]["sum"], 1)
        derived[fam] = x
    print(json.dumps({"note": "FETCH_SIZE x2 (gfx950) and KiB->bytes applied; derived ratios are unitless", "kernels": dict(top),
                      "derived": derived}, indent=1))


if __name__ == "__main__":
    main()
