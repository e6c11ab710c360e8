"""Per-op device time of the caption plans (HIP events around every op of an eager replay: omni_plan_profile), with the
algorithmic bytes / FLOPs of each op -> GB/s and TF/s per op, aggregated by (kernel family, tensor shape).
usage: python tools/caption_profile.py [capacity=128] [R=768] [repeat=2] [flags]   -> JSON on stdout, table on stderr
flags (comma separated): nomlp = stage-0 FFN as two launches; any boolean composition switch of Florence2Captioner to turn ON
(window_attn_v2, chan_apply_mfma: the candidate kernels)"""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    import torch
    from omniparser_amd import _lib as L
    from omniparser_amd.florence import Florence2Captioner
    from tools.make_weights import caption_dir, ensure_via_subprocess
    from omniparser_amd.opwork import op_kernel, op_shape, op_work
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 768
    rep = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    flags = sys.argv[4].split(",") if len(sys.argv) > 4 else []     # "nomlp": fc1 / fc2 of the C = 128 stage as two launches
    if "nomlp" in flags:
        Florence2Captioner.fuse_mlp = False
    for f in flags:                                                  # any other boolean composition switch of Florence2Captioner, "name" or "name=0"
        if f != "nomlp":
            name, _, val = f.partition("=")
            assert isinstance(vars(Florence2Captioner).get(name), bool), f"unknown composition switch {name!r}"
            setattr(Florence2Captioner, name, val != "0")
    ensure_via_subprocess("caption", seed=0)
    cap = Florence2Captioner(caption_dir(0), "cuda", precision="f32", resolution=R)
    cap.use_graph = False           # eager plans: every op is timed on its own
    cp = cap.plans(B, R, 20)
    g = torch.Generator().manual_seed(0)
    with torch.inference_mode(), torch.cuda.stream(cap.stream):
        # smooth inputs (what crops look like), not randn
        low = torch.randn(B, 8, 8, 3, generator=g).permute(0, 3, 1, 2)
        x = torch.nn.functional.interpolate(low, size=(R, R), mode="bicubic").permute(0, 2, 3, 1).contiguous()
        cp.x_in.t[:, :, :, :3] = x.to(cap.device)
        cp.reset()
        cp.encode_plan.run(cap.stream); cp.step_plan.run(cap.stream)
        cap.stream.synchronize()
    out = {"capacity": B, "R": R, "use_dma": bool(cp.use_dma), "plans": {}}
    for pname, plan, reps in (("encode", cp.encode_plan, rep), ("step", cp.step_plan, 20)):
        ms = [0.0] * len(plan.ops)
        with torch.inference_mode():
            if pname == "step":
                with torch.cuda.stream(cap.stream):
                    cp.reset()
            for _ in range(reps):
                for j, t in enumerate(plan.profile(cap.stream)):
                    ms[j] += t / reps
        agg = {}
        for op, t in zip(plan.ops, ms):
            f, b = op_work(op)
            key = (op_kernel(op), op_shape(op))
            a = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
            a[0] += 1; a[1] += t; a[2] += f; a[3] += b
        rows = [{"kernel": k[0], "shape": list(k[1]), "n": a[0], "ms": round(a[1], 3), "GBps": round(a[3] / max(a[1], 1e-9) / 1e6, 1),
                 "TFps": round(a[2] / max(a[1], 1e-9) / 1e9, 1)} for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])]
        fam = {}
        for r in rows:
            fam[r["kernel"]] = round(fam.get(r["kernel"], 0.0) + r["ms"], 3)
        out["plans"][pname] = {"total_ms": round(sum(ms), 3), "by_family_ms": dict(sorted(fam.items(), key=lambda kv: -kv[1])), "rows": rows}
        print(f"--- {pname}: {sum(ms):.2f} ms ({'per replay' if pname == 'encode' else 'per step'})", file=sys.stderr)
        for r in rows[:40]:
            print("%-26s %-22s n=%-3d %8.3f ms %8.1f GB/s %7.1f TF/s" % (r["kernel"], r["shape"], r["n"], r["ms"], r["GBps"], r["TFps"]), file=sys.stderr)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
