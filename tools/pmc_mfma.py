"""MFMA-pipe utilisation and sustained clock per kernel from two PMC summaries of tools/pmc_summary.py (separate rocprofv3
--kernel-trace --pmc passes over the same command): SQ counters (SQ_VALU_MFMA_BUSY_CYCLES ...) and GRBM_GUI_ACTIVE.
    python tools/pmc_mfma.py <pmc_summary_sq.json> <pmc_summary_grbm.json> > profiles/rN_mfma_utilisation.json
clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (the counter is summed over the 8 XCDs);
MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x per-XCD active cycles)."""
import json
import sys


def main():
    sq = json.load(open(sys.argv[1]))["kernels"]
    gr = json.load(open(sys.argv[2]))["kernels"]
    out = {"note": __doc__.split("\n\n")[0].replace("\n", " ") if False else
           "clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration; MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x per-XCD active cycles); "
           "LDS conflict share = SQ_LDS_BANK_CONFLICT / SQ_WAVE_CYCLES; two rocprofv3 --kernel-trace --pmc passes over `python tools/caption_profile.py 128 768 1`",
           "kernels": {}}
    rows = []
    for name, g in gr.items():
        ga = g.get("GRBM_GUI_ACTIVE")
        s = sq.get(name)
        if not ga or not s or "SQ_VALU_MFMA_BUSY_CYCLES" not in s:
            continue
        per_xcd = ga["sum"] / 8.0
        ms = ga["total_ms"]
        rec = {"launches": ga["launches"], "ms_per_launch": round(ms / max(ga["launches"], 1), 3),
               "clock_GHz": round(per_xcd / (ms * 1e-3) / 1e9, 3) if ms else None,
               "mfma_pipe_busy": round(s["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / (1024.0 * per_xcd), 3) if per_xcd else None}
        if "SQ_LDS_BANK_CONFLICT" in s and "SQ_WAVE_CYCLES" in s and s["SQ_WAVE_CYCLES"]["sum"]:
            rec["lds_conflict_share"] = round(s["SQ_LDS_BANK_CONFLICT"]["sum"] / s["SQ_WAVE_CYCLES"]["sum"], 4)
        rows.append((ms, name, rec))
    for ms, name, rec in sorted(rows, reverse=True)[:24]:
        out["kernels"][name] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
