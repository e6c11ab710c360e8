"""Per-op HIP-event timing of the detector plan: which layers the time goes to (writes gpurun_out/)."""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    import torch
    from omniparser_amd import _lib as L
    from omniparser_amd.util.yolov9 import YOLOv9Detector
    from tools.make_weights import ensure_blob
    prec = os.environ.get("OMNI_PRECISION", "f32")
    imgsz = 640 if os.environ.get("IMGSZ", "640") == "640" else (1080, 1920)
    batch = int(os.environ.get("BATCH", "1"))
    os.environ["OMNI_HIPGRAPH"] = "0"
    det = YOLOv9Detector(model_path=ensure_blob(0, 1, 1.0), device="cuda", precision=prec)
    dp = det.get_plan(1920, 1080, imgsz, 0.05, 0.1, 300, batch=batch)
    dp.plan.run(det.stream); det.stream.synchronize()
    rows = []
    names = {1: "conv", 2: "avgpool", 3: "maxpool", 4: "resize", 5: "letterbox", 6: "decode", 7: "nms"}
    for idx, op in enumerate(dp.plan.ops):
        p = L.Plan([op])
        p.run(det.stream); det.stream.synchronize()
        ms = p.time(10, det.stream)
        r = {"idx": idx, "kind": names.get(op.kind, op.kind), "us": round(ms * 1000, 2)}
        if op.kind == L.OP_CONV:
            i = op.i
            M, N, K = i[0] * i[10] * i[11], i[12], i[6] * i[7] * i[3]
            r.update(M=M, N=N, K=K, k=i[6], s=i[8], gflop=round(2 * M * N * K / 1e9, 3),
                     tflops=round(2 * M * N * K / (ms * 1e-3) / 1e12, 2))
        rows.append(r)
    tot = sum(r["us"] for r in rows)
    conv = [r for r in rows if r["kind"] == "conv"]
    print(f"precision {prec} imgsz {imgsz} batch {batch}: sum of per-op times {tot/1000:.3f} ms; conv {sum(r['us'] for r in conv)/1000:.3f} ms "
          f"({len(conv)} launches); other {sum(r['us'] for r in rows if r['kind']!='conv')/1000:.3f} ms")
    # aggregate by shape
    agg = {}
    for r in conv:
        key = (r["M"], r["N"], r["K"], r["k"], r["s"])
        a = agg.setdefault(key, {"n": 0, "us": 0.0, "gflop": 0.0})
        a["n"] += 1; a["us"] += r["us"]; a["gflop"] += r["gflop"]
    print("   M      N     K  k s   n   total_us  TF/s")
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1]["us"])[:40]:
        print("%6d %5d %5d %2d %d %3d %9.1f %6.1f" % (*key, a["n"], a["us"], a["gflop"] / a["us"] * 1e3))
    for r in rows:
        if r["kind"] != "conv":
            pass
    other = {}
    for r in rows:
        if r["kind"] != "conv":
            other.setdefault(r["kind"], [0, 0.0]); other[r["kind"]][0] += 1; other[r["kind"]][1] += r["us"]
    print("non-conv:", {k: (v[0], round(v[1], 1)) for k, v in other.items()})
    od = ROOT / "gpurun_out"; od.mkdir(exist_ok=True)
    (od / f"profile_plan_{prec}_{'640' if imgsz == 640 else 'native'}_b{batch}.json").write_text(json.dumps(rows))


if __name__ == "__main__":
    main()
