"""Name-independent import of a YOLOv9-E TorchScript blob (ref:util/yolov9.py:50 loads `icon_detect_v3/model.pt`, an opaque
`torch.jit` export whose attribute names are unknown here — the checkpoint is not on this box).

The product never executes the blob; it needs the blob's TENSORS under the canonical names `yolo_graph.py` lowers from.
Instead of trusting attribute names, the blob's own graph says which tensor plays which role:

  1. `blob.inlined_graph` is walked in program order; every `aten::_convolution` (+ the `aten::batch_norm` consuming its
     output, if any) becomes one *unit* with its hyper-parameters (stride, padding, groups) and the qualified attribute
     paths of its tensors (resolved through the `prim::GetAttr` chains back to `self`);
  2. `canonical_units()` lists the units of the YOLOv9-E inference graph in forward order (auxiliary
     branch a1..a9, CBLinear routers r10..r14, main branch b15..b28 with CBFuse sums, neck n29..n41, DDetect heads) with
     the hyper-parameters each must have;
  3. the two sequences are aligned.  Equal length + equal signatures is the normal case (any renaming / re-registration
     order / class renaming of the exporting module leaves the program order untouched); where a position disagrees, a
     unit with the wanted signature is looked up in a short window ahead (exporters that evaluate two sibling branches
     in the other order, e.g. the box and class branches of a head).  A DFL expressed as a fixed 1x1 conv is recognised
     and skipped (the decode kernel computes the expectation itself);
  4. BatchNorm may already be folded into the conv by the exporter (conv with bias, no batch_norm node): the unit then
     carries `.conv.bias` and no `.bn.*`, which `YoloV9EGraph.fold` accepts.

Anything that cannot be aligned raises `BlobImportError` naming the first offending unit of both sequences — a blob of a
different architecture fails at construction with a readable message, not with a KeyError deep inside the lowering.
"""
from typing import Dict, Iterator, List, Optional, Tuple

import torch


class BlobImportError(RuntimeError):
    pass


# ------------------------------------------------------------------------------------------ canonical graph (forward order)
def _cb(p, k, s=1, g=1):            # Conv2d(bias=False) + BatchNorm + SiLU
    yield {"kind": "cb", "name": p, "k": k, "s": s, "g": g}


def _plain(p, k=1, g=1):            # nn.Conv2d with bias, no norm
    yield {"kind": "c", "name": p, "k": k, "s": 1, "g": g}


def _rep(p):
    yield from _cb(p + ".conv1", 3)
    yield from _cb(p + ".conv2", 1)


def _csp(p, n=2):
    yield from _cb(p + ".cv1", 1)
    for i in range(n):
        yield from _rep(f"{p}.m.{i}.cv1")
        yield from _cb(f"{p}.m.{i}.cv2", 3)
    yield from _cb(p + ".cv2", 1)
    yield from _cb(p + ".cv3", 1)


def _elan(p):
    yield from _cb(p + ".cv1", 1)
    yield from _csp(p + ".cv2.0")
    yield from _cb(p + ".cv2.1", 3)
    yield from _csp(p + ".cv3.0")
    yield from _cb(p + ".cv3.1", 3)
    yield from _cb(p + ".cv4", 1)


def _adown(p):
    yield from _cb(p + ".cv1", 3, 2)
    yield from _cb(p + ".cv2", 1)


def canonical_units() -> Iterator[dict]:
    """Units of the YOLOv9-E inference graph in forward order (names = yolo_graph.py's canonical prefixes)."""
    yield from _cb("a1", 3, 2)
    yield from _cb("a2", 3, 2)
    yield from _elan("a3")
    for ad, el in (("a4", "a5"), ("a6", "a7"), ("a8", "a9")):
        yield from _adown(ad)
        yield from _elan(el)
    for r in ("r10", "r11", "r12", "r13", "r14"):
        yield from _plain(r + ".conv")
    yield from _cb("b15", 3, 2)
    yield from _cb("b17", 3, 2)
    yield from _elan("b19")
    for ad, el in (("b20", "b22"), ("b23", "b25"), ("b26", "b28")):
        yield from _adown(ad)
        yield from _elan(el)
    yield from _cb("n29.cv1", 1)
    yield from _cb("n29.cv5", 1)
    yield from _elan("n32")
    yield from _elan("n35")
    yield from _adown("n36")
    yield from _elan("n38")
    yield from _adown("n39")
    yield from _elan("n41")
    for i in range(3):                       # per scale: class branch and box branch (grouped convs, 4 groups); either may come first
        yield from _cb(f"head.cv3.{i}.0", 3)
        yield from _cb(f"head.cv3.{i}.1", 3)
        yield from _plain(f"head.cv3.{i}.2")
        yield from _cb(f"head.cv2.{i}.0", 3)
        yield from _cb(f"head.cv2.{i}.1", 3, 1, 4)
        yield from _plain(f"head.cv2.{i}.2", 1, 4)


N_HEAD_UNITS = 18


# ------------------------------------------------------------------------------------------ blob graph -> units
def _qualname(v) -> Optional[str]:
    """attribute path of a value produced by a prim::GetAttr chain rooted at the module's `self`."""
    parts = []
    n = v.node()
    while n.kind() == "prim::GetAttr":
        parts.append(n.s("name"))
        n = next(iter(n.inputs())).node()
    if n.kind() != "prim::Param":
        return None
    return ".".join(reversed(parts))


def _const(v):
    n = v.node()
    if n.kind() == "prim::Constant":
        return n.output().toIValue()
    if n.kind() == "prim::ListConstruct":
        return [_const(i) for i in n.inputs()]
    return None


def blob_units(blob) -> List[dict]:
    graph = blob.inlined_graph
    units, by_out = [], {}
    for n in graph.nodes():
        kind = n.kind()
        if kind in ("aten::_convolution", "aten::conv2d"):
            ins = list(n.inputs())
            w, b = _qualname(ins[1]), _qualname(ins[2])
            if w is None:
                raise BlobImportError(f"convolution weight is not a module attribute: {n}")
            stride, pad = _const(ins[3]), _const(ins[4])
            groups = _const(ins[8] if kind == "aten::_convolution" else ins[6])
            u = {"w": w, "b": b, "bn": None, "s": int(stride[0]), "p": int(pad[0]), "g": int(groups)}
            units.append(u)
            by_out[n.output().unique()] = u
        elif kind == "aten::batch_norm":
            ins = list(n.inputs())
            u = by_out.get(ins[0].unique())
            if u is None:
                raise BlobImportError("batch_norm that does not follow a convolution: unsupported export")
            names = [_qualname(v) for v in ins[1:5]]
            if any(x is None for x in names):
                raise BlobImportError("batch_norm statistics are not module attributes")
            u["bn"] = tuple(names) + (float(_const(ins[7])),)
    return units


# ------------------------------------------------------------------------------------------ alignment
def _sig_ok(c: dict, u: dict, w: torch.Tensor) -> bool:
    k = w.shape[-1]
    if w.dim() != 4 or k != c["k"] or u["s"] != c["s"] or u["g"] != c["g"] or u["p"] != c["k"] // 2:
        return False
    if c["kind"] == "c":
        return u["bn"] is None and u["b"] is not None
    return u["bn"] is not None or u["b"] is not None        # Conv+BN, or BN already folded into a biased conv


def _is_dfl(u: dict, w: torch.Tensor) -> bool:
    if w.dim() != 4 or tuple(w.shape[:1] + w.shape[2:]) != (1, 1, 1) or u["bn"] is not None:
        return False
    n = w.shape[1]
    return bool(torch.equal(w.flatten().float(), torch.arange(n, dtype=torch.float32)))


def import_state_dict(blob, window: int = 2) -> Dict[str, torch.Tensor]:
    """TorchScript YOLOv9-E blob -> canonical state dict {prefix.conv.weight, prefix.bn.*, ...} (f32, CPU)."""
    sd = {k: v.detach().float().cpu() for k, v in blob.state_dict().items()}
    units = [u for u in blob_units(blob)]
    for u in units:
        if u["w"] not in sd:
            raise BlobImportError(f"graph references tensor '{u['w']}' that is not in the blob's state_dict")
    units = [u for u in units if not _is_dfl(u, sd[u["w"]])]
    canon = list(canonical_units())
    if len(units) != len(canon):
        raise BlobImportError(f"the blob has {len(units)} convolution units, YOLOv9-E has {len(canon)}: not the architecture "
                              f"ref:util/yolov9.py expects (first blob unit: {units[0]['w'] if units else None})")
    out: Dict[str, torch.Tensor] = {}

    def take(c, u):
        name = c["name"]
        if c["kind"] == "c":
            out[name + ".weight"] = sd[u["w"]]
            out[name + ".bias"] = sd[u["b"]]
            return
        out[name + ".conv.weight"] = sd[u["w"]]
        if u["bn"] is not None:
            gw, gb, mu, var, eps = u["bn"]
            out[name + ".bn.weight"], out[name + ".bn.bias"] = sd[gw], sd[gb]
            out[name + ".bn.running_mean"], out[name + ".bn.running_var"] = sd[mu], sd[var]
            out[name + ".bn.eps"] = torch.tensor(eps, dtype=torch.float64)
        if u["b"] is not None:
            out[name + ".conv.bias"] = sd[u["b"]]

    def describe(u):
        return None if u is None else (u["w"], tuple(sd[u["w"]].shape), "s=%d g=%d bn=%s" % (u["s"], u["g"], u["bn"] is not None))

    # body: program order, with a short look-ahead for exporters that evaluate two differently shaped siblings the other way
    nb = len(canon) - N_HEAD_UNITS
    used = [False] * nb
    pos = 0
    for ci, c in enumerate(canon[:nb]):
        while pos < nb and used[pos]:
            pos += 1
        pick = next((j for j in range(pos, min(pos + window, nb)) if not used[j] and _sig_ok(c, units[j], sd[units[j]["w"]])), None)
        if pick is None:
            raise BlobImportError(f"cannot align unit #{ci} '{c['name']}' (k={c['k']} s={c['s']} g={c['g']} {c['kind']}) with the blob: "
                                  f"next unit is {describe(units[pos] if pos < nb else None)}")
        used[pick] = True
        take(c, units[pick])
    # heads: six 3-unit chains, per scale one class chain and one box chain (the box chain ends in a 4-group conv) in either order
    chains = [units[nb + 3 * t: nb + 3 * t + 3] for t in range(6)]
    for i in range(3):
        pair = chains[2 * i: 2 * i + 2]
        box = [ch for ch in pair if ch[2]["g"] == 4 and ch[1]["g"] == 4]
        cls = [ch for ch in pair if ch[2]["g"] == 1 and ch[1]["g"] == 1]
        if len(box) != 1 or len(cls) != 1:
            raise BlobImportError(f"head of scale {i}: expected one class chain and one 4-group box chain, found {[describe(ch[2]) for ch in pair]}")
        for role, ch in (("cv3", cls[0]), ("cv2", box[0])):
            cs = [c for c in canon[nb:] if c["name"].startswith(f"head.{role}.{i}.")]
            for c, u in zip(cs, ch):
                if not _sig_ok(c, u, sd[u["w"]]):
                    raise BlobImportError(f"head unit '{c['name']}' does not match {describe(u)}")
                take(c, u)
    check_shapes(out)
    return out


def check_shapes(sd: Dict[str, torch.Tensor]):
    """channel bookkeeping of the aligned units (the lowering would otherwise fail later with an opaque assert)."""
    def cout(p):
        return sd[p + ".conv.weight"].shape[0]
    def cin(p):
        return sd[p + ".conv.weight"].shape[1]
    problems = []
    for a, b in (("a1", "a2"), ("b15", "b17")):
        if cout(a) != cin(b):
            problems.append(f"{a} -> {b}: {cout(a)} output vs {cin(b)} input channels")
    if cin("a1") != 3 or cin("b15") != 3:
        problems.append("stems must take 3 input channels")
    for i in range(3):
        if sd[f"head.cv2.{i}.2.weight"].shape[0] != 64:
            problems.append(f"head.cv2.{i}.2: expected 4 x 16 DFL bins, found {sd[f'head.cv2.{i}.2.weight'].shape[0]} channels")
    nc = {sd[f"head.cv3.{i}.2.weight"].shape[0] for i in range(3)}
    if len(nc) != 1:
        problems.append(f"class heads disagree on the number of classes: {sorted(nc)}")
    if problems:
        raise BlobImportError("blob aligned with YOLOv9-E but its shapes are inconsistent: " + "; ".join(problems))


# ------------------------------------------------------------------------------------------ load-time proof
def verify_against_blob(blob, run_network, nc: int, size: int = 64, tol: float = 2e-2) -> float:
    """The import is PROVEN per checkpoint, not assumed: the blob itself is executed once on the CPU on a small seeded input
    (load time only — never on the inference path) and must agree with the lowered network on the same input.
    `run_network(x_nchw) -> [(cls_logits [1,nc,h,w], box_logits [1,64,h,w])] x 3` runs the HIP plan.  Returns the largest head
    difference relative to the head's own magnitude; raises BlobImportError above `tol` (a mis-assigned sibling shows up as
    O(1); rounding differences of a 300-layer net on an out-of-distribution probe stay orders of magnitude below)."""
    g = torch.Generator().manual_seed(1234)
    coarse = torch.rand(1, 3, size // 8, size // 8, generator=g)                 # blocky, GUI-like probe (flat 8x8 patches)
    x = coarse.repeat_interleave(8, 2).repeat_interleave(8, 3).contiguous()
    try:
        # the reference always runs the blob with the TorchScript optimiser off (ref:util/yolov9.py:120-121)
        with torch.inference_mode(), torch.jit.optimized_execution(False):
            ref = list(blob(x))
    except BlobImportError:
        raise
    except Exception as e:      # noqa: BLE001 — a blob that cannot run a 64x64 probe is an import failure, not a constructor crash
        raise BlobImportError(f"the blob failed on the {size}x{size} probe input: {type(e).__name__}: {e}") from e
    if len(ref) < 6:
        raise BlobImportError(f"the blob returned {len(ref)} tensors; ref:util/yolov9.py:92-96 consumes 6 ([cls, dist] per stride)")
    got = run_network(x)
    worst = 0.0
    proj = torch.arange(16, dtype=torch.float32).view(1, 1, 16, 1, 1)
    for i, (cls, box) in enumerate(got):
        rc, rd = ref[2 * i].float(), ref[2 * i + 1].float()
        if tuple(rc.shape) != tuple(cls.shape) or rc.shape[1] != nc:
            raise BlobImportError(f"stride #{i}: class map {tuple(cls.shape)} vs the blob's {tuple(rc.shape)}")
        b, _, h, w = box.shape
        dist = (box.view(b, 4, 16, h, w).softmax(2) * proj).sum(2)          # DFL expectation, as the blob does inside
        if tuple(rd.shape) != tuple(dist.shape):
            raise BlobImportError(f"stride #{i}: the blob's box output {tuple(rd.shape)} is not a DFL-reduced [1,4,h,w] map")
        worst = max(worst, (cls - rc).abs().max().item() / max(rc.abs().max().item(), 1e-3),
                    (dist - rd).abs().max().item() / max(rd.abs().max().item(), 1e-3))
    if not worst <= tol:
        raise BlobImportError(f"imported network disagrees with the blob on a probe input (largest relative head difference {worst:.3e} > {tol}): "
                              "tensor roles were mis-assigned or the blob is not YOLOv9-E")
    return worst
