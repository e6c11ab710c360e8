"""Run every GPU parity check without stopping at the first failure; write gpurun_out/selftest.json."""
import json
import os
import sys
import time
import traceback
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))


def main():
    import torch
    import gpu_checks as G
    from omniparser_amd import _lib as L
    out = {"device": torch.cuda.get_device_name(0), "nproc": os.cpu_count()}
    checks = [] if "--only-extra" in sys.argv else [
        ("mfma_layout", lambda: G.check_mfma_layout()),
        ("conv_f32", lambda: G.check_conv(L.F32)),
        ("conv_f16", lambda: G.check_conv(L.F16)),
        ("pools_f32", lambda: G.check_pools(L.F32)),
        ("pools_f16", lambda: G.check_pools(L.F16)),
        ("letterbox_f32", lambda: G.check_letterbox(L.F32)),
        ("post_nc1", lambda: G.check_post(seed=0, nc=1, frac=0.08)),
        ("post_nc3", lambda: G.check_post(seed=1, nc=3, frac=0.05)),
        ("post_dense", lambda: G.check_post(seed=2, nc=1, frac=0.6)),
        ("nms_known", lambda: G.check_nms_known_answers()),
    ]
    if "--caption" in sys.argv:
        checks.append(("caption_ops_f32", lambda: G.check_caption_ops(L.F32)))
        checks.append(("caption_ops_f16", lambda: G.check_caption_ops(L.F16)))
        checks.append(("captioner_r64", lambda: G.check_captioner(R=64, n=5)[0]))
        if "--r768" in sys.argv:
            checks.append(("captioner_r768", lambda: G.check_captioner(R=768, n=2)[0]))
    if "--e2e" in sys.argv:
        checks.append(("e2e_r64", lambda: G.check_end_to_end(width=0.5, R=64, image_seed=1)))
    if "--detector" in sys.argv:
        w = float(os.environ.get("SELFTEST_WIDTH", "0.5"))
        checks.append(("detector_f32", lambda: G.check_detector(width=w, image_seeds=(0, 1))[0]))
        checks.append(("detector_f16", lambda: G.check_detector(width=w, image_seeds=(0,), precision="f16", with_f64=False)[0]))
    for name, fn in checks:
        t = time.time()
        try:
            out[name] = {"ok": True, "result": fn()}
        except Exception as e:   # noqa
            out[name] = {"ok": False, "error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()[-1500:]}
        out[name]["sec"] = round(time.time() - t, 2)
        print(name, "OK" if out[name]["ok"] else "FAIL " + out[name]["error"][:300], flush=True)
    od = ROOT / "gpurun_out"
    od.mkdir(exist_ok=True)
    (od / "selftest.json").write_text(json.dumps(out, indent=1, default=str))
    print("failed:", [k for k, v in out.items() if isinstance(v, dict) and not v.get("ok", True)])


if __name__ == "__main__":
    main()
