"""Why does omni_captioner_caption differ from Florence2Captioner.caption_crops on the MI355X (tests/test_gpu_i_model_capi.py, session 5)?
GPU-vs-GPU bisection: graph / eager on both sides, determinism of each side, then x_in -> vision_out -> enc_out -> logits of the first
micro-batch bit for bit.  Prints one JSON line per finding."""
import ctypes
import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from omniparser_amd import _lib as L                       # noqa: E402
from omniparser_amd import bundle as B                      # noqa: E402
from omniparser_amd.florence import Florence2Captioner      # noqa: E402
from omniparser_amd.synth import synthetic_screenshot       # noqa: E402
from tools.make_weights import ensure_caption_checkpoint    # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def dev_read(ptr, nbytes, dtype):
    out = np.empty(nbytes, np.uint8)
    rc = hip.hipMemcpy(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(nbytes), 2)
    assert rc == 0, rc
    return out.view(dtype)


def say(**kw):
    print(json.dumps(kw), flush=True)


def rows_equal(a, b):
    return [bool(np.array_equal(a[r], b[r])) for r in range(a.shape[0])]


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    capn = 8
    img = synthetic_screenshot(0, 1920, 1080)
    rng = np.random.default_rng(5)
    rects = []
    for _ in range(11):
        x0, y0 = int(rng.integers(0, 1800)), int(rng.integers(0, 1000))
        rects.append([x0, y0, x0 + int(rng.integers(8, 110)), y0 + int(rng.integers(8, 70))])
    tmp = Path(tempfile.mkdtemp())
    cap = Florence2Captioner(ensure_caption_checkpoint(0), "cuda", precision="f32", resolution=R)
    dimg = torch.from_numpy(img).cuda()
    py = [cap.caption_crops(dimg, rects, max_new_tokens=20, batch_size=capn).numpy().astype(np.int32) for _ in range(3)]
    say(what="python graph path deterministic", ok=[bool(np.array_equal(py[0], p)) for p in py[1:]])
    # python, first micro-batch only, keep the intermediate tensors
    py8 = cap.caption_crops(dimg, rects[:capn], max_new_tokens=20, batch_size=capn).numpy().astype(np.int32)
    cp = cap.plans(capn, R, 20)
    torch.cuda.synchronize()
    taps_py = {k: v.detach().cpu().numpy().copy() for k, v in
               (("x_in", cp.x_in.t), ("vision_out", cp.vision_out.t), ("enc_out", cp.enc_out.t), ("logits", cp.logits.t), ("ids", cp.ids))}
    say(what="python rows 0..7 equal between 11-crop and 8-crop call", rows=rows_equal(py[0][:capn], py8))
    B.export_captioner(cap, tmp / "cap.omniplan", capacity=capn, max_new_tokens=20)
    for mode in ("graph", "eager"):
        if mode == "eager":
            os.environ["OMNI_HIPGRAPH"] = "0"
        c = L.CModel(tmp / "cap.omniplan", "captioner")
        os.environ.pop("OMNI_HIPGRAPH", None)
        cid = [c.caption(img, rects) for _ in range(3)]
        T = py[0].shape[1]
        say(what=f"C {mode} deterministic", ok=[bool(np.array_equal(cid[0], x)) for x in cid[1:]])
        say(what=f"C {mode} vs python, per row", rows=rows_equal(cid[0][:, :T], py[0]))
        c8 = c.caption(img, rects[:capn])
        say(what=f"C {mode} 8-crop vs python 8-crop, per row", rows=rows_equal(c8[:, :T], py8))
        for name, ref in taps_py.items():
            ptr, nb = c.tensor(name)
            got = dev_read(ptr, nb, ref.dtype).reshape(ref.shape)
            n = capn
            if name == "x_in" or ref.dtype != np.float32:
                say(what=f"C {mode} tap {name}", bitwise_rows=rows_equal(got[:n], ref[:n]))
            else:
                g, r = got[:n].reshape(n, -1).astype(np.float64), ref[:n].reshape(n, -1).astype(np.float64)
                say(what=f"C {mode} tap {name}", bitwise_rows=rows_equal(got[:n], ref[:n]),
                    rel_err=[float(np.abs(g[k] - r[k]).max() / (np.abs(r[k]).max() + 1e-30)) for k in range(n)],
                    finite=[bool(np.isfinite(g).all()), bool(np.isfinite(r).all())])
        c.close()
    # python eager
    cap2 = Florence2Captioner(ensure_caption_checkpoint(0), "cuda", precision="f32", resolution=R)
    cap2.use_graph = False
    pe = cap2.caption_crops(dimg, rects, max_new_tokens=20, batch_size=capn).numpy().astype(np.int32)
    say(what="python eager vs python graph, per row", rows=rows_equal(pe, py[0]), shapes=[list(pe.shape), list(py[0].shape)])
    # margins of the python logits at the last step (how decisive is the arg-max?)
    lg = taps_py["logits"].reshape(capn, -1)
    top2 = np.sort(lg, axis=1)[:, -2:]
    say(what="python last-step top1-top2 margins (rows 0..7)", margins=[float(t[1] - t[0]) for t in top2])


if __name__ == "__main__":
    main()
