"""Name-independent TorchScript blob import (omniparser_amd/yolo_import.py): the real `icon_detect_v3/model.pt` is not on
this box, so the importer is exercised with blobs whose attribute names, registration order, class names, sibling evaluation
order and BatchNorm folding differ from the oracle's — the tensors must land on the same canonical roles every time."""
import copy

import pytest
import torch
import torch.nn as nn

from omniparser_amd.yolo_import import BlobImportError, canonical_units, import_state_dict

WIDTH = 0.25


@pytest.fixture(scope="module")
def base():
    from oracle.yolov9e_ref import build_random_detector
    return build_random_detector(seed=3, nc=2, width=WIDTH)


def _trace(model, tmp_path, name):
    with torch.no_grad():
        ts = torch.jit.trace(model, torch.rand(1, 3, 64, 64), check_trace=False)
    p = tmp_path / name
    ts.save(str(p))
    return torch.jit.load(str(p), map_location="cpu").eval()


def _canonical(model):
    return {k: v.float() for k, v in model.state_dict().items() if "num_batches" not in k}


def _assert_same(sd, model):
    ref = _canonical(model)
    for k, v in ref.items():
        assert k in sd, k
        assert torch.equal(sd[k], v), k
    extra = [k for k in sd if k not in ref and not k.endswith(".bn.eps")]
    assert not extra, extra[:5]


def mangle(model):
    """Rename every child attribute, reverse the registration order and rename the classes; forwards keep working through
    properties that alias the old names (TorchScript records the REGISTERED names)."""
    for m in list(model.modules()):
        kids = list(m._modules.items())
        if not kids or isinstance(m, (nn.Sequential, nn.ModuleList)):
            continue
        alias = {}
        m._modules.clear()
        for i, (old, child) in enumerate(reversed(kids)):
            new = f"z{len(kids) - i}_{old[::-1]}"
            m._modules[new] = child
            alias[old] = new
        props = {old: property(lambda self, n=new: self._modules[n]) for old, new in alias.items()}
        m.__class__ = type("Blk" + m.__class__.__name__[::-1], (m.__class__,), props)
    return model


def test_canonical_unit_count_matches_the_public_architecture():
    units = list(canonical_units())
    assert len(units) == 309                                   # fused-RepConv count would be 261
    assert sum(u["kind"] == "c" for u in units) == 5 + 6 and len({u["name"] for u in units}) == 309


def test_import_oracle_named_blob(base, tmp_path):
    _assert_same(import_state_dict(_trace(base, tmp_path, "plain.pt")), base)


def test_import_renamed_reordered_blob(base, tmp_path):
    blob = _trace(mangle(copy.deepcopy(base)), tmp_path, "mangled.pt")
    names = list(blob.state_dict().keys())
    assert not any(n.startswith(("a1.", "head.", "b15.")) for n in names), names[:3]      # nothing of the oracle's naming survives
    _assert_same(import_state_dict(blob), base)


def test_import_upstream_style_names(base, tmp_path):
    """public YOLOv9 exports name their layers model.N.* (N = yaml index); same program order."""
    class Up(nn.Module):
        def __init__(self, m):
            super().__init__()
            order = ["a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "r10", "r11", "r12", "r13", "r14", "b15", "f16", "b17", "f18",
                     "b19", "b20", "f21", "b22", "b23", "f24", "b25", "b26", "f27", "b28", "n29", "n32", "n35", "n36", "n38", "n39", "n41", "head"]
            self.idx = {n: i + 1 for i, n in enumerate(order)}
            self.model = nn.ModuleList([nn.Identity()] + [getattr(m, n) for n in order])
            self.inner = [m]                                   # not registered: only its forward is borrowed

        def forward(self, x):
            m = self.inner[0]
            saved = {n: getattr(m, n) for n in self.idx}
            try:
                for n, i in self.idx.items():
                    m._modules[n] = self.model[i]
                return m.forward(x)
            finally:
                for n, v in saved.items():
                    m._modules[n] = v
    blob = _trace(Up(copy.deepcopy(base)), tmp_path, "upstream.pt")
    assert all(k.startswith("model.") for k in blob.state_dict())
    _assert_same(import_state_dict(blob), base)


def test_import_box_branch_first_and_dfl_conv(base, tmp_path):
    """public DDetect evaluates the box branch before the class branch and applies DFL as a fixed 1x1 conv."""
    m = copy.deepcopy(base)
    head = m.head

    class Dfl(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = nn.Conv2d(16, 1, 1, bias=False)
            self.conv.weight.data[:] = torch.arange(16, dtype=torch.float32).view(1, 16, 1, 1)

        def forward(self, x):
            b, _, h, w = x.shape
            return self.conv(x.view(b, 4, 16, h * w).transpose(2, 1).softmax(1)).view(b, 4, h, w)
    head.dfl_mod = Dfl()

    def fwd(feats):
        out = []
        for i, f in enumerate(feats):
            box = head.dfl_mod(head.cv2[i](f))
            out.append(head.cv3[i](f))
            out.append(box)
        return out
    head.forward = fwd
    _assert_same(import_state_dict(_trace(m, tmp_path, "boxfirst.pt")), base)


def test_import_bn_folded_blob_lowers_to_the_same_weights(base, tmp_path):
    from omniparser_amd.planner import PlanBuilder
    from omniparser_amd.yolo_graph import YoloV9EGraph
    from omniparser_amd import _lib as L
    from oracle.yolov9e_ref import Conv
    m = copy.deepcopy(base)
    for mod in m.modules():
        if isinstance(mod, Conv):
            inv = mod.bn.weight / torch.sqrt(mod.bn.running_var + mod.bn.eps)
            fused = nn.Conv2d(mod.conv.in_channels, mod.conv.out_channels, mod.conv.kernel_size, mod.conv.stride, mod.conv.padding,
                              groups=mod.conv.groups, bias=True)
            fused.weight.data = mod.conv.weight.data * inv.view(-1, 1, 1, 1)
            fused.bias.data = mod.bn.bias.data - mod.bn.running_mean * inv
            mod.conv, mod.bn = fused, nn.Identity()
    sd_f = import_state_dict(_trace(m, tmp_path, "fused.pt"))
    sd_u = import_state_dict(_trace(base, tmp_path, "unfused.pt"))
    assert not any(".bn." in k for k in sd_f)
    gf = YoloV9EGraph(sd_f, PlanBuilder("cpu", L.F32), 1, 64, 64)
    gu = YoloV9EGraph(sd_u, PlanBuilder("cpu", L.F32), 1, 64, 64)
    for u in canonical_units():
        if u["kind"] == "cb":
            (wf, bf), (wu, bu) = gf.fold(u["name"]), gu.fold(u["name"])
            assert torch.allclose(wf, wu, rtol=0, atol=1e-6) and torch.allclose(bf, bu, rtol=0, atol=1e-6), u["name"]


def test_foreign_architecture_fails_with_a_readable_message(tmp_path):
    class Tiny(nn.Module):
        def __init__(self):
            super().__init__()
            self.c = nn.Conv2d(3, 8, 3, 2, 1)

        def forward(self, x):
            return [self.c(x)] * 6
    with pytest.raises(BlobImportError, match="convolution units"):
        import_state_dict(_trace(Tiny(), tmp_path, "tiny.pt"))


def test_standin_blob_is_rebuilt_bit_for_bit_from_committed_calibration(tmp_path):
    """tools/make_weights.py: seeded initialisation + the committed calibration constants (tools/standin_calibration/*.pt) give the
    same TorchScript blob state on every box — the frame lists of the parity tests (EXACT_FRAMES / WELL_FRAMES) were scanned on exactly
    these weights.  The constants hold what the calibration computes and nothing else."""
    import torch
    from oracle.yolov9e_ref import YOLOv9E, calibrated_keys
    from tools import make_weights as MW
    for seed, nc, width in ((0, 1, 0.25), (0, 1, 0.5), (0, 1, 1.0), (1, 2, 0.25)):
        assert MW.calibration_path(seed, nc, width).exists(), (seed, nc, width)
    calib = torch.load(str(MW.calibration_path(0, 1, 0.25)), map_location="cpu")
    want = set(calibrated_keys(YOLOv9E(nc=1, width=0.25).state_dict()))
    assert set(calib) == want | {"margin", "pass_rate"} and len(want) == 900
    out = tmp_path / "model.pt"
    MW.make_blob(out, seed=0, nc=1, width=0.25)
    a = torch.jit.load(str(out), map_location="cpu").state_dict()
    b = torch.jit.load(str(MW.ensure_blob(seed=0, nc=1, width=0.25)), map_location="cpu").state_dict()
    assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)
