"""Per-shape tile / split-K choice for the GEMMs of the captioner's DECODE step (conv_split_kernel at M = the decode plan's rows), measured
IN CONTEXT on the MI355X — the decode-side twin of tools/conv_autotune.py.

A decode step is ~110 launch-bound kernels; 37 of them are GEMMs over 128-512 rows whose tile / split-K the launcher's throughput
heuristic picks (>= 512 workgroups: 64x64 tiles and 6 K splits of four slices for [352 x 768 x 768], plus a reduce launch each).  For every
row count a merged decode plan can have (multiples of 32, florence.py::decode_bucket) this tool measures every (tile, split-K) combination
per shape inside an eager replay of the step plan (HIP events around each op), picks the fastest, checks the tuned step as a captured
graph, and writes omniparser_amd/decode_tuning_gfx950.json (loaded by florence.py::decode_tuning).

  python tools/decode_autotune.py [--write] [rows ...]      (GPU box)"""
import ctypes
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
TILES = (1, 2, 3)
SPLITS = (1, 2, 3, 4, 6, 8, 12)


def main():
    import torch
    from omniparser_amd import _lib as L
    from omniparser_amd import florence as FL
    from omniparser_amd.planner import conv_key
    from tools.make_weights import ensure_caption_checkpoint
    rows_list = [int(a) for a in sys.argv[1:] if a.isdigit()] or list(range(128, 513, 32))
    FL._DECODE_TUNING = None                                     # build with the launcher's heuristic
    cap = FL.Florence2Captioner(ensure_caption_checkpoint(0), "cuda", precision="f32", resolution=768)
    cap.use_graph = False
    out = {"rows": {}, "choices": {}}

    def clone(op, tile, splits):
        c = L.OmniOp()
        ctypes.memmove(ctypes.byref(c), ctypes.byref(op), ctypes.sizeof(c))
        if c.kind == L.OP_CONV and c.i[20] == 1:
            c.i[22], c.i[23] = tile, splits
        return c

    def key_of(op):
        i = op.i
        return conv_key(i[0] * i[10] * i[11], i[12], i[6] * i[7] * i[3], i[6], i[8])

    def graph_ms(dec, ops, reps=6):
        """ms of the 20 steps of one decode as graph replays.  The step plan is STATEFUL (a device step counter indexes the id table and
        the self-attention cache: T = 21 positions), so every measured round starts from `reset()` and replays exactly 20 steps — the
        first version of this tool replayed 120 steps in a row and wrote beyond the cache (a GPU memory fault, session 12)."""
        p = L.Plan(ops)
        with torch.cuda.stream(cap.stream):
            dec.reset()
        p.run(cap.stream); cap.stream.synchronize()
        p.capture(cap.stream)
        best = 1e9
        for _ in range(reps):
            with torch.cuda.stream(cap.stream):
                dec.reset()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cap.stream)
                for _ in range(20):
                    p.replay(cap.stream)
                e1.record(cap.stream)
            cap.stream.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    for rows in rows_list:
        dec = FL._DecodePlans(cap, rows, 768, 20)
        for kv in dec.cross_kv:
            kv.t.normal_()
        ops = list(dec.step_plan.ops)
        tunable = [j for j, op in enumerate(ops) if op.kind == L.OP_CONV and op.i[20] == 1]
        combos = [(0, 0)] + [(t, s) for t in TILES for s in SPLITS]
        per = {}
        for (t, s) in combos:
            p = L.Plan([clone(op, t, s) for op in ops])
            with torch.cuda.stream(cap.stream):
                dec.reset()
            p.run(cap.stream); cap.stream.synchronize()
            acc = None
            for _ in range(4):                                    # (5 steps since the reset above: inside the 21-position state)
                ms = p.profile(cap.stream)
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
            if bt < 0.97 * base:
                picked[k] = {"choice": list(best), "n": len(idxs), "heuristic_us": round(1000 * base, 1), "tuned_us": round(1000 * bt, 1)}
        tuned = [clone(op, *(picked.get(key_of(op), {"choice": (0, 0)})["choice"] if (op.kind == L.OP_CONV and op.i[20] == 1) else (0, 0))) for op in ops]
        g0, g1 = graph_ms(dec, ops), graph_ms(dec, tuned)
        out["rows"][rows] = {"ops": len(ops), "tunable": len(tunable), "graph_ms_heuristic": round(g0, 4), "graph_ms_tuned": round(g1, 4), "picked": picked}
        if g1 < 0.985 * g0:
            for k, v in picked.items():
                out["choices"][k] = v["choice"]
        del dec
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    print(json.dumps(out))
    if "--write" in sys.argv:
        table = json.dumps({"what": "tile code (1 = 64x64, 2 = 128x64, 3 = 128x128) and split-K count per GEMM shape MxNxKk1s1 of the captioner's decode "
                                    "step (M = rows of the merged decode plan, multiples of 32); measured by tools/decode_autotune.py on an MI355X "
                                    "(in-context HIP-event times, adopted per row count only when the captured step graph got faster); absent shapes "
                                    "use the launcher's heuristic", "choices": out["choices"]}, indent=0, sort_keys=True)
        (ROOT / "omniparser_amd" / "decode_tuning_gfx950.json").write_text(table)
        d = ROOT / "gpurun_out" / "r6_autotune"
        d.mkdir(parents=True, exist_ok=True)
        (d / "decode_tuning_gfx950.json").write_text(table)


if __name__ == "__main__":
    main()
