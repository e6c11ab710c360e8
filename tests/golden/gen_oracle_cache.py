"""Fill tests/golden/oracle_cache/ in the CPU container: the CAPTION-ORACLE halves of the `-m gpu` end-to-end tests (transformers
Florence-2 fp32 on the oracle's own crop tensors — 3-4 s of 128 threads per 768x768 crop on the GPU box, which the lease should not pay).

  python tests/golden/gen_oracle_cache.py [bench] [refimgs] [e2e] [tiled] [stream] [verify]        (default: all; resumable: rows already cached are hits)

Each section computes, with the ORACLE only (no device), the crop rectangles the corresponding GPU test will send to
`gpu_checks._OracleCaptioner.caption_crops` and calls it with OMNI_ORACLE_CACHE_WRITE=1, so the rows land in the committed cache file
(tests/oracle_cache.py: keyed by the sha256 of each crop's pixel tensor inside a file named after the oracle model's digest — a GPU test
whose device crops differ from the oracle's simply misses and computes live).  Records what it did in oracle_cache/MANIFEST.json (stand-in
versions, blob digest, row counts per section)."""
import hashlib
import json
import os
import sys
import time
from pathlib import Path
from types import SimpleNamespace

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ["OMNI_ORACLE_CACHE_WRITE"] = "1"


def sha16(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:16]


def main():
    import torch
    from PIL import Image
    import gpu_checks as G
    import oracle_cache as OC
    from oracle import detector_ref as D
    from oracle import tiling_ref as TR
    from omniparser_amd.pipeline import ScreenParser
    from omniparser_amd.synth import BENCH_SEEDS, synthetic_ocr, synthetic_screenshot
    from omniparser_amd.util import utils as U
    from tools.make_weights import CAPTION_STANDIN, DETECTOR_STANDIN, build_random_captioner, ensure_blob, ensure_caption_checkpoint
    want = set(sys.argv[1:]) or {"bench", "refimgs", "e2e", "tiled", "stream", "verify"}
    model = build_random_captioner(0)
    cdir = ensure_caption_checkpoint(0)
    proc = U.FlorenceProcessor(cdir)
    manifest_p = OC.CACHE_DIR / "MANIFEST.json"
    manifest = json.loads(manifest_p.read_text()) if manifest_p.exists() else {}
    manifest.update(detector_standin=DETECTOR_STANDIN, caption_standin=CAPTION_STANDIN, oracle_model_digest=OC.model_digest(model), sections=manifest.get("sections", {}))
    glue = SimpleNamespace(iou_threshold=0.7)
    kw = dict(BOX_TRESHOLD=0.05, output_coord_in_ratio=True, use_local_semantics=True, iou_threshold=0.7, scale_img=False, batch_size=128)

    def section(name, fn):
        if name not in want:
            return
        t0 = time.time()
        before = dict(OC.STATS)
        info = fn() or {}
        info.update(seconds=round(time.time() - t0, 1), rows_computed=OC.STATS["misses"] - before["misses"], rows_already_cached=OC.STATS["hits"] - before["hits"])
        manifest["sections"][name] = info
        OC.CACHE_DIR.mkdir(parents=True, exist_ok=True)
        manifest_p.write_text(json.dumps(manifest, indent=1, sort_keys=True))
        print(name, info, flush=True)

    def bench():
        # tests/test_gpu_z_bench_path.py::test_bench_path_parity_batch8_full_width_r768 (check_bench_path, R = 768, the benched frames)
        blob = ensure_blob(0, 1, 1.0)
        cpu_model = torch.jit.load(str(blob), map_location="cpu").eval()
        imgs = [synthetic_screenshot(s, 1920, 1080) for s in BENCH_SEEDS]
        crops = []
        for s, im in zip(BENCH_SEEDS, imgs):
            rb, rs, rc = D.predict(cpu_model, Image.fromarray(im), conf=0.05, imgsz=640, iou=0.1, max_det=300)
            texts, obox = synthetic_ocr(s, 1920, 1080, 40)
            crops.append(ScreenParser.glue(glue, rb, 1920, 1080, obox, texts)[1])
        # round 6: EVERY crop of the benched batch (345 rows), not the seams only — check_bench_path compares them all
        ocap = G._OracleCaptioner(model, 768)
        for f in range(len(imgs)):
            ocap.caption_crops(imgs[f], crops[f], max_new_tokens=20, batch_size=8)
        return {"blob_sha16": sha16(blob), "seeds": list(BENCH_SEEDS), "crops_per_frame": [len(c) for c in crops], "rows": sum(len(c) for c in crops)}

    def verify():
        # staleness audit: N rows of the committed file recomputed LIVE on the benched frames (use_cache=False) and compared with what the
        # file holds for the same key — ids equal, margins within 1e-4 (thread count changes the last bits of a CPU matmul)
        import numpy as np
        from oracle import preprocess_ref as PR
        from omniparser_amd.florence import CLIP_MEAN, CLIP_STD
        blob = ensure_blob(0, 1, 1.0)
        cpu_model = torch.jit.load(str(blob), map_location="cpu").eval()
        cache = OC.CaptionCache(model)
        n, bad, dm = 0, [], 0.0
        for s in BENCH_SEEDS[:2]:
            im = synthetic_screenshot(s, 1920, 1080)
            rb, rs, rc = D.predict(cpu_model, Image.fromarray(im), conf=0.05, imgsz=640, iou=0.1, max_det=300)
            texts, obox = synthetic_ocr(s, 1920, 1080, 40)
            cr = ScreenParser.glue(glue, rb, 1920, 1080, obox, texts)[1]
            pick = [k for k in range(len(cr)) if cache.rows.get(OC.row_key(PR.caption_pixel_values(im, cr[k], 768, CLIP_MEAN, CLIP_STD), 768, 20))][:6]
            live = G._OracleCaptioner(model, 768, use_cache=False)
            ref = live.caption_crops(im, [cr[k] for k in pick], max_new_tokens=20, batch_size=8)
            for j, k in enumerate(pick):
                rec = cache.rows[OC.row_key(PR.caption_pixel_values(im, cr[k], 768, CLIP_MEAN, CLIP_STD), 768, 20)]
                ids = [int(v) for v in ref[j].tolist()][:len(rec["ids"])]
                n += 1
                if ids != rec["ids"] or any(int(v) != 1 for v in ref[j].tolist()[len(rec["ids"]):]):
                    bad.append((s, k))
                if rec["margin"] is not None:
                    dm = max(dm, abs(rec["margin"] - live.margins[j]))
        assert not bad and dm < 1e-4, (bad, dm)
        return {"rows_recomputed_live": n, "ids_equal": n - len(bad), "max_margin_diff": dm}

    def refimgs():
        # tests/test_gpu_j_reference_images.py (check_end_to_end, width 1.0, R = 768, first 8 crops) on the reference's own two images
        gold = json.loads((HERE / "reference_images.json").read_text())
        blob = ensure_blob(0, 1, 1.0)
        out = {}
        for name in ("word.png", "demo_image.jpg"):
            img = Image.open(HERE / "ref_imgs" / name)
            texts, obox = synthetic_ocr(gold["ocr"]["seed"], img.size[0], img.size[1], gold["ocr"]["n"])
            rb, el = G.oracle_end_to_end(img, blob, proc, 768, 8, dict(kw, ocr_bbox=obox, ocr_text=texts))
            out[name] = {"boxes": int(rb.shape[0]), "elements": len(el)}
        return {"blob_sha16": sha16(blob), "images": out}

    def e2e():
        # tests/test_gpu_d_pipeline.py::test_end_to_end_get_som_labeled_img (check_end_to_end, width 0.5, R = 64, seed 1: every crop)
        blob = ensure_blob(0, 1, 0.5)
        img = Image.fromarray(synthetic_screenshot(1, 1920, 1080))
        texts, obox = synthetic_ocr(1, 1920, 1080, 40)
        rb, el = G.oracle_end_to_end(img, blob, proc, 64, None, dict(kw, ocr_bbox=obox, ocr_text=texts))
        return {"blob_sha16": sha16(blob), "boxes": int(rb.shape[0]), "elements": len(el)}

    def tiled():
        # tests/test_gpu_d_pipeline.py::test_tiled_4k_end_to_end_captions_token_exact (check_tiled_captions, width 0.5, R = 64, seed 4)
        blob = ensure_blob(0, 1, 0.5)
        cpu_model = torch.jit.load(str(blob), map_location="cpu").eval()
        img = synthetic_screenshot(4, 3840, 2160)
        texts, obox = synthetic_ocr(4, 3840, 2160, 60)
        origins, tw, th = ScreenParser.tile_origins(3840, 2160)
        rb, rs, rc = TR.predict_tiled(cpu_model, img, origins, tw, th)
        crops = ScreenParser.glue(glue, rb, 3840, 2160, obox, texts)[1]
        G._OracleCaptioner(model, 64).caption_crops(img, crops, max_new_tokens=20, batch_size=64)
        # ... and at the crop size BASELINE configs[4] names (768x768, 64-crop micro-batches): test_tiled_4k_end_to_end_captions_token_exact_r768
        G._OracleCaptioner(model, 768).caption_crops(img, crops, max_new_tokens=20, batch_size=16)
        return {"blob_sha16": sha16(blob), "boxes": int(rb.shape[0]), "crops": len(crops)}

    def stream():
        # tests/test_gpu_k_stream_parity.py: frames at stream.RESOLUTION_MIX sizes, 64x64 crops, every crop (see gpu_checks.stream_parity_cases)
        info = {}
        blob = ensure_blob(0, 1, 1.0)
        cpu_model = torch.jit.load(str(blob), map_location="cpu").eval()
        first_size, n768 = tuple(G.stream_parity_cases()[0][1:]), 0
        for (seed, w, h) in G.stream_parity_cases():
            img = synthetic_screenshot(seed, w, h)
            texts, obox = synthetic_ocr(seed, w, h, 40)
            rb, rs, rc = D.predict(cpu_model, Image.fromarray(img), conf=0.05, imgsz=640, iou=0.1, max_det=300)
            crops = ScreenParser.glue(glue, rb, w, h, obox, texts)[1]
            G._OracleCaptioner(model, 64).caption_crops(img, crops, max_new_tokens=20, batch_size=64)
            info[f"{seed}:{w}x{h}"] = {"boxes": int(rb.shape[0]), "crops": len(crops)}
            if (w, h) == first_size and n768 < 2:           # test_stream_768_crops_first_frames_vs_oracle: the two frames of the first size
                G._OracleCaptioner(model, 768).caption_crops(img, crops, max_new_tokens=20, batch_size=8)
                n768 += 1
        return {"blob_sha16": sha16(blob), "frames": info}

    section("e2e", e2e)
    section("tiled", tiled)
    section("stream", stream)
    section("bench", bench)
    section("refimgs", refimgs)
    section("verify", verify)


if __name__ == "__main__":
    main()
