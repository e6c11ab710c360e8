"""Round-3 closing probe (information for the next round, no product change): time of the merged 20-step decode at 384 rows (the
benched decode plan: 345 crops rounded up to a multiple of 128) against 352 rows (a multiple of 32)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    import torch
    from omniparser_amd.florence import Florence2Captioner
    from tools.make_weights import caption_dir, ensure_via_subprocess
    ensure_via_subprocess("caption", seed=0)
    cap = Florence2Captioner(caption_dir(0), "cuda", precision="f32", resolution=768)
    out = {}
    for rows in (384, 352):
        dec = cap.decode_plans(rows, 768, 20)
        with torch.inference_mode(), torch.cuda.stream(cap.stream):
            for kv in dec.cross_kv:
                kv.t.normal_()
            best = 1e9
            for _ in range(4):
                dec.reset()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cap.stream)
                cap._decode_merged(dec, rows, 20)
                e1.record(cap.stream)
                cap.stream.synchronize()
                best = min(best, e0.elapsed_time(e1))
        out[f"decode20_rows{rows}_ms"] = round(best, 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
