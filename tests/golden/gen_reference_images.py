"""Golden fixtures on the REFERENCE'S OWN input images: /root/reference/imgs/word.png (1919x1079 RGBA — the odd size
exercises the int() truncation of ref:util/yolov9.py:77-78 and the RGBA -> RGB conversion of :68-71) and demo_image.jpg (3240x2160 =
BASELINE configs[0]).  Run here (the container that has /root/reference):

  * the two images are copied to tests/golden/ref_imgs/ as INPUT fixtures (data, not source);
  * the REFERENCE's own `YOLOv9Detector` (ref:util/yolov9.py, imported under the shims of gen_golden.py) runs `_preprocess` and
    `predict` on them with the repo's seeded stand-in blob (tools/make_weights.py, seed 0, nc 1, full width — the real
    icon_detect_v3/model.pt cannot be downloaded): letterbox geometry + a digest of the network input, final boxes / scores;
  * the REFERENCE's own `get_som_labeled_img` (ref:util/utils.py:417-496) runs on them with that detector, a synthetic OCR
    fixture and a duck-typed captioner that records the crops it is handed: element list (type / bbox / source / order) and the
    integer crop rectangles of ref:util/utils.py:95-102.

Consumers: tests/test_reference_images_cpu.py (oracle == these vectors, bit for bit) and tests/test_gpu_j_reference_images.py
(product vs oracle on the same images).  The stand-in blob is regenerated from its seed wherever the tests run; its digest is
recorded so that a blob that differs (other torch build) is reported as such instead of as a parity failure.
"""
import hashlib
import json
import shutil
import sys
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
REF = Path("/root/reference")
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(HERE))

IMAGES = ("word.png", "demo_image.jpg")
CONF, IMGSZ, NMS_IOU = 0.05, 640, 0.1          # ref:util/omniparser.py:30 (BOX_TRESHOLD in the shipped config is 0.05), ref:util/utils.py:431
OCR_SEED, OCR_N = 7, 40


def state_digest(blob_path) -> str:
    """digest of the stand-in's parameters (not of the file: TorchScript archives carry timestamps)"""
    m = torch.jit.load(str(blob_path), map_location="cpu")
    h = hashlib.sha256()
    for k, v in sorted(m.state_dict().items()):
        h.update(k.encode()); h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def main():
    import gen_golden
    gen_golden.install_shims()
    import util.utils as RU
    import util.yolov9 as RY
    from PIL import Image
    from omniparser_amd.synth import synthetic_ocr
    from tools.make_weights import ensure_blob
    (HERE / "ref_imgs").mkdir(exist_ok=True)
    blob = ensure_blob(seed=0, nc=1, width=1.0)
    det = RY.YOLOv9Detector(model_path=str(blob), device="cpu")
    out = {"blob": {"seed": 0, "nc": 1, "width": 1.0, "state_sha256": state_digest(blob)}, "conf": CONF, "imgsz": IMGSZ, "nms_iou": NMS_IOU,
           "ocr": {"seed": OCR_SEED, "n": OCR_N}, "images": {}}

    class RecCfg: name_or_path = "florence-recorder"; model_type = "florence2"

    class RecCap:          # duck-typed captioner: one id per crop (its index), crops recorded by the processor below
        config = RecCfg(); device = torch.device("cpu")
        def generate(self, input_ids=None, pixel_values=None, **k): return torch.arange(pixel_values.shape[0]).view(-1, 1)

    for name in IMAGES:
        shutil.copyfile(REF / "imgs" / name, HERE / "ref_imgs" / name)
        img = Image.open(HERE / "ref_imgs" / name)
        rec = {"size": list(img.size), "mode": img.mode, "file_sha256": hashlib.sha256((HERE / "ref_imgs" / name).read_bytes()).hexdigest()}
        rgb = RY.YOLOv9Detector._load_image(img)
        x, scale, pad_left, pad_top = det._preprocess(rgb, IMGSZ)
        rec["preprocess"] = {"shape": list(x.shape), "scale": scale, "pad_left": pad_left, "pad_top": pad_top,
                             "u8_sha256": hashlib.sha256((x[0] * 255.0).round().to(torch.uint8).numpy().tobytes()).hexdigest(),
                             "sum": float(x.double().sum())}
        r = det.predict(img, conf=CONF, imgsz=IMGSZ, iou=NMS_IOU)[0]
        rec["predict"] = {"boxes_bits": r.boxes.xyxy.contiguous().view(torch.int32).tolist(), "conf_bits": r.boxes.conf.contiguous().view(torch.int32).tolist(),
                          "n": int(r.boxes.xyxy.shape[0])}
        # the reference's get_som_labeled_img on this image: reference detector, synthetic OCR, recording captioner
        texts, obox = synthetic_ocr(OCR_SEED, img.size[0], img.size[1], OCR_N)
        crops = []

        class RecProc:
            def __call__(self, images=None, text=None, return_tensors="pt", **k):
                crops.extend([list(np.asarray(im).shape[:2]) for im in images])
                class B(dict):
                    def to(self, **k): return self
                return B({"input_ids": torch.zeros(len(images), 1, dtype=torch.long), "pixel_values": torch.zeros(len(images), 3, 8, 8)})
            def batch_decode(self, ids, skip_special_tokens=True): return [f" cap{int(i)} " for i in ids.view(-1)]

        rects = []
        real_resize = RU.cv2.resize

        def rec_resize(im, dsize, **k):
            rects.append([int(im.shape[0]), int(im.shape[1])])
            return real_resize(im, dsize, **k)
        RU.cv2.resize = rec_resize
        enc, lab, elems = RU.get_som_labeled_img(img, det, BOX_TRESHOLD=CONF, output_coord_in_ratio=True, ocr_bbox=obox, draw_bbox_config=None,
                                                 caption_model_processor={"model": RecCap(), "processor": RecProc()}, ocr_text=texts,
                                                 use_local_semantics=True, iou_threshold=0.7, scale_img=False, batch_size=128)
        RU.cv2.resize = real_resize
        rec["som"] = {"elems": elems, "label_keys": list(lab.keys()), "crop_shapes_hw": rects, "n_crops": len(rects)}
        out["images"][name] = rec
        print(name, img.size, img.mode, "boxes", rec["predict"]["n"], "elements", len(elems), "crops", len(rects))
    (HERE / "reference_images.json").write_text(json.dumps(out))
    print("wrote", HERE / "reference_images.json")


if __name__ == "__main__":
    main()
