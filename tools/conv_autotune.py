"""Per-shape tile / split-K choice for the detector's split-f16 convolutions, measured IN CONTEXT on the MI355X.

Round 5's kernel trace of a batch-1 detector replay (profiles/r5_s5_det_b1_kernel_stats.csv): 475 kernels back to back (mean gap 0.3 us)
of 9.4 us mean duration — the pass is bound by per-kernel latency, and 197 of the kernels are split-K reduce launches of ~5.5 us
(24 % of the pass).  The launcher's heuristic (>= 512-768 workgroups, then shrink tiles, then split K) was written for throughput.
This tool measures, for every conv of the plan, every (tile, split-K) combination in the plan's real sequence (HIP events around each
op of an eager replay: caches as the predecessors left them), picks the fastest per SHAPE, checks the tuned plan against the
untuned one as captured hipGraphs, and writes omniparser_amd/conv_tuning_gfx950.json (loaded by util/yolov9.py::conv_tuning).

  python tools/conv_autotune.py [--write]      (GPU box)   -> JSON summary on stdout; --write replaces the committed table"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
TILES = (1, 2, 3)                                   # 64x64, 128x64, 128x128
SPLITS = (1, 2, 3, 4, 6, 8, 12, 16, 24, 32)
PLANS = (("b1_640", 1, 640), ("b8_640", 8, 640), ("b1_native", 1, (1080, 1920)))


def main():
    import ctypes
    import torch
    from omniparser_amd import _lib as L
    from omniparser_amd.planner import conv_key
    from omniparser_amd.util import yolov9 as Y
    from tools.make_weights import ensure_blob
    Y._CONV_TUNING = None                                       # build the plans with the launcher's heuristic
    det = Y.YOLOv9Detector(model_path=ensure_blob(0, 1, 1.0), device="cuda", precision="f32")
    det.use_graph = False
    out = {"plans": {}, "choices": {}}
    choices = {}

    def clone(op, tile, splits):
        c = L.OmniOp()
        ctypes.memmove(ctypes.byref(c), ctypes.byref(op), ctypes.sizeof(c))
        if c.kind == L.OP_CONV and c.i[20] == 1:
            c.i[22], c.i[23] = tile, splits
        return c

    def key_of(op):
        i = op.i
        return conv_key(i[0] * i[10] * i[11], i[12], i[6] * i[7] * i[3], i[6], i[8])

    def graph_ms(ops, reps=30):
        p = L.Plan(ops)
        p.run(det.stream); det.stream.synchronize()
        p.capture(det.stream)
        p.time(3, det.stream)
        return min(p.time(reps, det.stream) for _ in range(3))

    for name, batch, imgsz in PLANS:
        dp = det.get_plan(1920, 1080, imgsz, 0.05, 0.1, 300, batch=batch)
        ops = list(dp.plan.ops)
        tunable = [j for j, op in enumerate(ops) if op.kind == L.OP_CONV and op.i[20] == 1]
        combos = [(0, 0)] + [(t, s) for t in TILES for s in SPLITS]
        per = {}                                                # (tile, splits) -> [ms per op]
        for (t, s) in combos:
            p = L.Plan([clone(op, t, s) for op in ops])
            p.run(det.stream); det.stream.synchronize()
            acc = None
            for _ in range(3):
                ms = p.profile(det.stream)
                acc = ms if acc is None else [min(a, b) for a, b in zip(acc, ms)]
            per[(t, s)] = acc
        by_shape = {}
        for j in tunable:
            by_shape.setdefault(key_of(ops[j]), []).append(j)
        picked = {}
        for k, idxs in by_shape.items():
            base = sum(per[(0, 0)][j] for j in idxs)
            best = min(combos, key=lambda c: sum(per[c][j] for j in idxs))
            bt = sum(per[best][j] for j in idxs)
            if bt < 0.97 * base:                                # a choice must beat the heuristic by more than the timing noise
                picked[k] = {"choice": list(best), "n": len(idxs), "heuristic_us": round(1000 * base, 1), "tuned_us": round(1000 * bt, 1)}
        tuned_ops = [clone(op, *(picked.get(key_of(op), {"choice": (0, 0)})["choice"] if (op.kind == L.OP_CONV and op.i[20] == 1) else (0, 0)))
                     for op in ops]
        g0, g1 = graph_ms(ops), graph_ms(tuned_ops)
        out["plans"][name] = {"ops": len(ops), "tunable_convs": len(tunable), "shapes": len(by_shape), "shapes_retuned": len(picked),
                              "graph_ms_heuristic": round(g0, 4), "graph_ms_tuned": round(g1, 4), "picked": picked}
        if g1 < 0.98 * g0:                                      # adopt a plan's choices only if the whole graph replay is faster
            for k, v in picked.items():
                choices[k] = v["choice"]
    out["choices"] = choices
    print(json.dumps(out))
    if "--write" in sys.argv:
        path = ROOT / "omniparser_amd" / "conv_tuning_gfx950.json"
        (ROOT / "gpurun_out" / "r5_autotune").mkdir(parents=True, exist_ok=True)
        table = json.dumps({"what": "tile code (1 = 64x64, 2 = 128x64, 3 = 128x128) and split-K count per conv shape MxNxKk<kernel>s<stride> of the "
                                            "detector's split-f16 convolutions; measured by tools/conv_autotune.py on an MI355X (in-context HIP-event "
                                            "times, adopted per plan only when the captured graph got faster); absent shapes use the launcher's heuristic",
                                    "choices": choices}, indent=0, sort_keys=True)
        path.write_text(table)
        (ROOT / "gpurun_out" / "r5_autotune" / "conv_tuning_gfx950.json").write_text(table)      # gpurun merges gpurun_out/ back


if __name__ == "__main__":
    main()
