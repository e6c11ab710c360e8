#!/usr/bin/env bash
# round-5 session 12 (evidence only): every 1088x1920 frame the a-priori scan passed (seeds 1, 5, 6, 7, 8, 12, 14 of profiles/r5_parity_frame_scan.md)
# through the detector vs the oracle with exact=True — how many of the scanned native frames does the MI355X reproduce box for box?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5_s12
timeout 600 python - > gpurun_out/r5_s12/native_exact.json 2> gpurun_out/r5_s12/native_exact.err <<'PY'
import json, sys
sys.path.insert(0, "tests")
import gpu_checks as G
seeds = (1, 5, 6, 7, 8, 12, 14)
out, det = G.check_detector(width=1.0, image_seeds=seeds, imgsz=(1080, 1920))
rows = []
for rec in out["images"]:
    try:
        G.assert_detector_frame(rec, exact=True)
        ok, why = True, ""
    except AssertionError as e:
        ok, why = False, str(e)[:200]
    rows.append({"seed": rec["seed"], "exact": ok, "n_ref": rec["n_ref"], "n_gpu": rec["n_gpu"], "cand_ref": rec["cand_ref"], "cand_gpu": rec["cand_gpu"],
                 "cand_borderline": rec.get("cand_borderline"), "near_ties": rec["near_ties"], "score_ties": rec["score_ties"],
                 "head_err_max": max(max(e) for e in rec["head_err(cls,dist)"]), "matched_min_iou": rec.get("matched_min_iou"),
                 "unmatched_boxes": rec.get("unmatched_boxes"), "why": why})
print(json.dumps({"frames": len(rows), "exact": sum(r["exact"] for r in rows), "rows": rows}))
PY
tail -c 2500 gpurun_out/r5_s12/native_exact.json; echo; grep -v Warning gpurun_out/r5_s12/native_exact.err | tail -3 | cut -c1-300
