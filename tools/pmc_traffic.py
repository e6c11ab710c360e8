"""Combine the two single-counter PMC summaries of tools/pmc_summary.py (FETCH_SIZE pass, WRITE_SIZE pass — separate rocprofv3 runs of the
same command) into the traffic record bench.py quotes in `roofline.traffic`:
    python tools/pmc_traffic.py <pmc_summary_FETCH_SIZE.json> <pmc_summary_WRITE_SIZE.json> <crop_encodes> <algorithmic_bytes_per_crop> > profiles/r3_pmc_traffic.json
crop_encodes = rows of caption-encode work the profiled process ran (tools/caption_profile.py 128 768 1: one eager pass when the plan is
built + one profiled pass = 256).  Per kernel family: HBM bytes fetched (FETCH_SIZE x 2 per the gfx950 note of MI355X_MICROARCH.md) and
written per launch, and the bandwidth over the kernel's own duration."""
import json
import sys

GEMM = ("gemm_dma_kernel", "mlp_fused_kernel", "conv_split_kernel", "conv_igemm_kernel", "splitk_reduce_kernel")


def main():
    f = json.load(open(sys.argv[1]))["kernels"]
    w = json.load(open(sys.argv[2]))["kernels"]
    crops = int(sys.argv[3])
    algo = float(sys.argv[4])
    fams = {}
    for name in sorted(set(f) | set(w)):
        fe, wr = f.get(name, {}).get("FETCH_SIZE"), w.get(name, {}).get("WRITE_SIZE")
        if not fe or not wr:
            continue
        ms = 0.5 * (fe["total_ms"] + wr["total_ms"])
        fams[name] = {"launches": fe["launches"], "fetch_bytes": fe["sum"], "write_bytes": wr["sum"], "ms": round(ms, 3),
                      "bytes_per_launch": round((fe["sum"] + wr["sum"]) / max(fe["launches"], 1)),
                      "TBps": round((fe["sum"] + wr["sum"]) / (ms * 1e-3) / 1e12, 3) if ms else None,
                      "fetch_over_write": round(fe["sum"] / max(wr["sum"], 1.0), 3)}
    g = {k: v for k, v in fams.items() if k.startswith(GEMM)}
    gf, gw, gl = sum(v["fetch_bytes"] for v in g.values()), sum(v["write_bytes"] for v in g.values()), sum(v["launches"] for v in g.values())
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over `python tools/caption_profile.py 128 768 1` "
                     "(FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md; the 25 kernels with the largest total duration)",
           "what_ran": f"{crops} crop-encodes at 768x768 (one eager pass when the 128-row plan is built + one profiled pass) and 21 decode steps over 128 rows",
           "gemm_kernels": ", ".join(GEMM), "gemm_launches": gl, "gemm_fetch_bytes": gf, "gemm_write_bytes": gw,
           "gemm_fetch_bytes_per_crop": round(gf / crops), "gemm_write_bytes_per_crop": round(gw / crops),
           "gemm_bytes_per_launch": round((gf + gw) / max(gl, 1)), "algorithmic_gemm_bytes_per_crop": round(algo),
           "ratio": round((gf + gw) / crops / algo, 3), "families": fams}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
