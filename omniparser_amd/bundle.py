"""Plan bundles: a detector / captioner plan exported as ONE file that the model-level C entry points of libomni_amd.so
(include/omni_amd.h: omni_detector_create / omni_detector_infer, omni_captioner_create / omni_captioner_caption) load without
Python or torch — the boundary SURVEY 8b proposes for non-Python hosts, behind the same header as the op / plan ABI.

The reference builds its models inside Python (ref:util/utils.py:63-77 get_yolo_model / get_caption_model_processor, transformers
and a TorchScript blob underneath); here the Python graph builders (yolo_import / yolo_graph / florence) stay the only place that
knows the architectures, and what they produce — an immutable op list over a fixed set of device buffers — is what travels:

  header   "OMNIPLN<abi>", counts
  tensors  size, role (scratch / zero-initialised / constant), file offset of the constant data
  plans    name + ops; every device pointer of an op as (tensor index, byte offset)
  named    I/O and state tensors by name (tensor index, byte offset, size)
  ints     scalars the entry points need (sizes, token ids, float bit patterns)
  data     constants (packed weights, tables), 256-byte aligned

The loader allocates the tensors, uploads the constants, rebuilds the ops with the new addresses and captures each plan as a
hipGraph.  Export is an offline step (like building an inference engine file); results through the C entry points are
bit-identical to the Python objects' (tests/test_gpu_i_model_capi.py, tests/test_model_capi_emu_cpu.py).
"""
import bisect
import struct
from typing import Dict

import numpy as np
import torch

from . import _lib as L
from . import planner

MAGIC = b"OMNIPLN%d" % L.ABI_VERSION      # the last character is the ABI the ops were written under: omni_model_load rejects any other
ROLE = {"scratch": 0, "zero": 1, "const": 2}


def _name(s: str) -> bytes:
    b = s.encode()
    assert len(b) < 32, s
    return b.ljust(32, b"\0")


def f32_bits(x: float) -> int:
    return struct.unpack("<i", struct.pack("<f", float(x)))[0]


def write_bundle(path, plans: Dict[str, list], named: Dict[str, tuple], ints: Dict[str, int]):
    """plans: name -> list of OmniOp; named: name -> (tensor, byte offset, nbytes); ints: name -> int."""
    live = planner.live_tensors()
    bases = [r[0] for r in live]

    def resolve(ptr):
        k = bisect.bisect_right(bases, ptr) - 1
        if k < 0 or ptr >= live[k][0] + live[k][1]:      # one past the end belongs to whatever lies behind, never to this tensor
            raise ValueError(f"device pointer {ptr:#x} does not belong to a PlanBuilder tensor: the plan cannot be exported")
        return k, ptr - live[k][0]

    used, order = {}, []

    def tid(k):
        if k not in used:
            used[k] = len(order)
            order.append(k)
        return used[k]

    plan_blobs = []
    for pname, ops in plans.items():
        recs = []
        for op in ops:
            ptrs = []
            for j in range(8):
                v = op.p[j]
                if not v:
                    ptrs.append((-1, 0))
                else:
                    k, off = resolve(int(v))
                    ptrs.append((tid(k), off))
            recs.append((int(op.kind), int(op.dtype), ptrs, [int(x) for x in op.i], [float(x) for x in op.f]))
        plan_blobs.append((pname, recs))
    named_recs = []
    for nm, (t, off, nbytes) in named.items():
        k, o = resolve(t.data_ptr() + off)
        named_recs.append((nm, tid(k), o, int(nbytes)))
    # layout
    head = struct.pack("<8sIIII", MAGIC, len(order), len(plan_blobs), len(named_recs), len(ints))
    tens_size = 24 * len(order)
    plans_size = sum(32 + 4 + len(recs) * (8 + 8 * 12 + 32 * 4 + 8 * 4) for _, recs in plan_blobs)
    named_size = len(named_recs) * (32 + 4 + 8 + 8)
    ints_size = len(ints) * (32 + 8)
    off = len(head) + tens_size + plans_size + named_size + ints_size
    data_off = []
    for k in order:
        _, nbytes, role, _t = live[k]
        if role == "const":
            off = (off + 255) // 256 * 256
            data_off.append(off)
            off += nbytes
        else:
            data_off.append(0)
    with open(path, "wb") as f:
        f.write(head)
        for k, do in zip(order, data_off):
            _, nbytes, role, _t = live[k]
            f.write(struct.pack("<QIIQ", nbytes, ROLE[role], 0, do))
        for pname, recs in plan_blobs:
            f.write(_name(pname) + struct.pack("<I", len(recs)))
            for kind, dtype, ptrs, ii, ff in recs:
                f.write(struct.pack("<ii", kind, dtype))
                for t, o in ptrs:
                    f.write(struct.pack("<iq", t, o))
                f.write(struct.pack("<32i", *ii))
                f.write(struct.pack("<8f", *ff))
        for nm, t, o, nb in named_recs:
            f.write(_name(nm) + struct.pack("<iqq", t, o, nb))
        for nm, v in ints.items():
            f.write(_name(nm) + struct.pack("<q", int(v)))
        for k, do in zip(order, data_off):
            _, nbytes, role, t = live[k]
            if role == "const":
                f.seek(do)
                f.write(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes())
    return {"tensors": len(order), "const_bytes": sum(live[k][1] for k in order if live[k][2] == "const"),
            "device_bytes": sum(live[k][1] for k in order), "ops": {p: len(r) for p, r in plan_blobs}}


def read_bundle(path):
    """Python reader of the format (tests; the product reader is csrc/model_api.hip)."""
    b = open(path, "rb").read()
    magic, nt, npl, nn, ni = struct.unpack_from("<8sIIII", b, 0)
    assert magic == MAGIC
    p = 24
    tensors = []
    for _ in range(nt):
        tensors.append(struct.unpack_from("<QIIQ", b, p)); p += 24
    plans = {}
    for _ in range(npl):
        nm = b[p:p + 32].rstrip(b"\0").decode(); p += 32
        (n,) = struct.unpack_from("<I", b, p); p += 4
        ops = []
        for _ in range(n):
            kind, dtype = struct.unpack_from("<ii", b, p); p += 8
            ptrs = [struct.unpack_from("<iq", b, p + 12 * j) for j in range(8)]; p += 96
            ii = struct.unpack_from("<32i", b, p); p += 128
            ff = struct.unpack_from("<8f", b, p); p += 32
            ops.append((kind, dtype, ptrs, ii, ff))
        plans[nm] = ops
    named = {}
    for _ in range(nn):
        nm = b[p:p + 32].rstrip(b"\0").decode(); p += 32
        named[nm] = struct.unpack_from("<iqq", b, p); p += 20
    ints = {}
    for _ in range(ni):
        nm = b[p:p + 32].rstrip(b"\0").decode(); p += 32
        (ints[nm],) = struct.unpack_from("<q", b, p); p += 8
    return {"tensors": tensors, "plans": plans, "named": named, "ints": ints}


# ------------------------------------------------------------------------------------------ the two models
def export_detector(det, iw: int, ih: int, path, imgsz=640, conf=0.05, iou=0.1, max_det=300, batch=1):
    """One detector configuration (image size, network size, thresholds, batch) -> bundle for omni_detector_create.
    ref:util/yolov9.py:115-136 is what omni_detector_infer stands behind."""
    dp = det.get_plan(iw, ih, imgsz, conf, iou, max_det, batch=batch)
    t = lambda x: (x, 0, x.numel() * x.element_size())
    named = {"img": t(dp.img), "out_boxes": t(dp.out_boxes), "out_scores": t(dp.out_scores), "out_cls": t(dp.out_cls),
             "out_count": t(dp.out_count)}
    ints = {"model": 1, "batch": batch, "img_w": iw, "img_h": ih, "max_det": max_det, "dtype": det.dtype}
    return write_bundle(path, {"detect": dp.plan.ops}, named, ints)


def export_captioner(cap, path, capacity=8, max_new_tokens=20):
    """One caption plan set (capacity rows at the captioner's crop resolution) -> bundle for omni_captioner_create.
    ref:util/utils.py:88-132 (crop, resize, processor, generate) is what omni_captioner_caption stands behind."""
    from .florence import CLIP_MEAN, CLIP_STD
    R = cap.resolution
    cp = cap.plans(capacity, R, max_new_tokens)
    pb = planner.PlanBuilder(cap.device, cap.dtype)          # run-time tables of the crop op, registered like plan tensors
    lut = pb.upload(torch.from_numpy((np.arange(256).astype(np.float64) * (1 / 255)).astype(np.float32)))
    boxes = pb.raw((capacity, 4), torch.int32)
    c64 = pb.raw((capacity, 64, 64, 3), torch.uint8)
    t = lambda x: (x, 0, x.numel() * x.element_size())
    named = {"x_in": t(cp.x_in.t), "ids": t(cp.ids), "finished": t(cp.finished), "step": t(cp.step), "lut": t(lut), "boxes": t(boxes),
             "c64": t(c64),
             # read-only taps for hosts that want the intermediate results (omni_model_tensor): DaViT output, encoder output, last logits
             "vision_out": t(cp.vision_out.t), "enc_out": t(cp.enc_out.t), "logits": t(cp.logits.t)}
    ks = 0
    if R != 64:
        b, k = L.resample_coeffs(64, R, 1)
        bb, kk = pb.upload(torch.from_numpy(b)), pb.upload(torch.from_numpy(k))
        tmp = pb.raw((capacity, 64, R, 3), torch.uint8)
        named.update(bic_bounds=t(bb), bic_coef=t(kk), tmp=t(tmp))
        ks = k.shape[1]
    cp._bundle_keep = pb.keep
    ints = {"model": 2, "capacity": capacity, "R": R, "T": cp.T, "max_new": max_new_tokens, "start_token": cp.start_token, "pad": cap.w.pad,
            "eos": cap.w.eos, "ksize": ks, "ldo": cp.x_in.ld, "dtype": cap.dtype}
    for j in range(3):
        ints[f"mean{j}"] = f32_bits(CLIP_MEAN[j])
        ints[f"std{j}"] = f32_bits(CLIP_STD[j])
    return write_bundle(path, {"encode": cp.encode_plan.ops, "step": cp.step_plan.ops}, named, ints)
