"""YOLOv9-E as a static op list for the gfx950 plan executor.

What the reference runs as an opaque TorchScript module (ref:util/yolov9.py:50,121) is lowered here,
from the module's state_dict, to ~190 launches of the implicit-GEMM conv kernel plus pooling /
nearest-resize kernels (topology: SURVEY.md Appendix B):

  * BatchNorm folded into conv weight + f32 bias; RepConvN (3x3 (+) 1x1) re-parameterised to one 3x3;
    grouped head convs expanded to block-diagonal dense convs (tiny layers, keeps one GEMM kernel);
  * sibling 1x1 convs on the same input (RepNCSP cv1|cv2, head box|cls stems) merged into one GEMM;
  * chunk / concat / split are channel-slice views (no copies); CBFuse = nearest-resize accumulate
    chain into a scratch tensor consumed as the residual operand of the producing conv's epilogue;
  * box logits stay 4x16 DFL bins — the expectation is fused into the decode kernel.

The state dict arrives under canonical names from `yolo_import.import_state_dict`, which assigns the blob's tensors to
their roles from the blob's own graph (program order + hyper-parameters), not from its attribute names.
"""
from typing import Dict, List

import torch

from . import _lib as L
from .planner import PlanBuilder, View


class YoloV9EGraph:
    def __init__(self, state_dict: Dict[str, torch.Tensor], pb: PlanBuilder, B: int, TH: int, TW: int,
                 wcache: Dict = None):
        self.sd = state_dict
        self.pb = pb
        self.B, self.TH, self.TW = B, TH, TW
        self.nc = self.sd["head.cv3.0.2.weight"].shape[0]
        # packed device weights are shared by every plan built for this detector
        self.wcache = wcache if wcache is not None else {}

    def packed(self, key, make, cin_pad=None):
        """(packed weight on device, f32 bias on device) for `key`, built once via make() -> (W, b)."""
        ck = (key, self.pb.dtype, cin_pad, str(self.pb.device))
        if ck not in self.wcache:
            w, b = make()
            wp = self.pb.pack_weight(w.float(), cin_pad)
            bp = self.pb.upload(b.detach().float()) if b is not None else None
            self.wcache[ck] = (wp, bp)
        return self.wcache[ck]

    def conv(self, key, make, x, out, k, s=1, act=L.ACT_SILU, res=None, cin_pad=None):
        wp, bp = self.packed(key, make, cin_pad)
        return self.pb.conv(x, wp, bp, out, k, s, act=act, res=res)

    # ------------------------------------------------------------ weight transforms
    def fold(self, prefix):
        """Conv2d(bias=False)+BN(eps) -> (W, b); a blob exported with BN already folded (conv with bias, no norm) passes through."""
        w = self.sd[prefix + ".conv.weight"]
        if prefix + ".bn.weight" not in self.sd:
            if prefix + ".conv.bias" not in self.sd:
                raise KeyError(f"{prefix}: neither BatchNorm statistics nor a conv bias in the imported blob")
            return w, self.sd[prefix + ".conv.bias"]
        g, b = self.sd[prefix + ".bn.weight"], self.sd[prefix + ".bn.bias"]
        mu, var = self.sd[prefix + ".bn.running_mean"], self.sd[prefix + ".bn.running_var"]
        eps = float(self.sd[prefix + ".bn.eps"]) if prefix + ".bn.eps" in self.sd else 1e-3
        inv = g / torch.sqrt(var + eps)
        bias = b - mu * inv
        if prefix + ".conv.bias" in self.sd:
            bias = bias + self.sd[prefix + ".conv.bias"] * inv
        return w * inv.view(-1, 1, 1, 1), bias

    def fold_rep(self, prefix):
        """RepConvN: conv1 (3x3, BN) + conv2 (1x1, BN) -> one 3x3."""
        w3, b3 = self.fold(prefix + ".conv1")
        w1, b1 = self.fold(prefix + ".conv2")
        w = w3.clone()
        w[:, :, 1:2, 1:2] += w1
        return w, b3 + b1

    @staticmethod
    def dense_from_groups(w, groups):
        """[Cout, Cin/g, k, k] grouped -> block-diagonal dense [Cout, Cin, k, k]."""
        if groups == 1:
            return w
        cout, cig, kh, kw = w.shape
        cog = cout // groups
        d = w.new_zeros(cout, cig * groups, kh, kw)
        for g in range(groups):
            d[g * cog:(g + 1) * cog, g * cig:(g + 1) * cig] = w[g * cog:(g + 1) * cog]
        return d

    def cout(self, prefix):
        return self.sd[prefix + ".conv.weight"].shape[0]

    # ------------------------------------------------------------ blocks
    def conv_bn(self, prefix, x: View, out: View = None, k=1, s=1, res: View = None, cin_pad=None) -> View:
        if out is None:
            p = k // 2
            out = self.pb.alloc(x.B, (x.H + 2 * p - k) // s + 1, (x.W + 2 * p - k) // s + 1, self.cout(prefix))
        return self.conv(prefix, lambda: self.fold(prefix), x, out, k, s, res=res, cin_pad=cin_pad)

    def repncsp(self, prefix, x: View, out: View, n=2) -> View:
        pb = self.pb
        c_ = self.cout(prefix + ".cv1")
        cat = pb.alloc(x.B, x.H, x.W, 2 * c_)

        def merged():
            w1, b1 = self.fold(prefix + ".cv1")
            w2, b2 = self.fold(prefix + ".cv2")
            return torch.cat([w1, w2], 0), torch.cat([b1, b2], 0)
        self.conv(prefix + ".cv1|cv2", merged, x, cat, 1)
        cur = cat.slice(0, c_)
        tmp_a = pb.alloc(x.B, x.H, x.W, c_)
        for i in range(n):
            self.conv(f"{prefix}.m.{i}.cv1", lambda i=i: self.fold_rep(f"{prefix}.m.{i}.cv1"), cur, tmp_a, 3)
            dst = cat.slice(0, c_) if i == n - 1 else pb.alloc(x.B, x.H, x.W, c_)
            self.conv(f"{prefix}.m.{i}.cv2", lambda i=i: self.fold(f"{prefix}.m.{i}.cv2"), tmp_a, dst, 3, res=cur)
            cur = dst
        return self.conv_bn(prefix + ".cv3", cat, out, 1)

    def elan(self, prefix, x: View, out: View = None, res: View = None) -> View:
        pb = self.pb
        c3 = self.cout(prefix + ".cv1")
        c4 = self.cout(prefix + ".cv2.1")
        c2 = self.cout(prefix + ".cv4")
        cat = pb.alloc(x.B, x.H, x.W, c3 + 2 * c4)
        self.conv_bn(prefix + ".cv1", x, cat.slice(0, c3), 1)
        y1 = cat.slice(c3 // 2, c3 // 2)
        t = pb.alloc(x.B, x.H, x.W, c4)
        self.repncsp(prefix + ".cv2.0", y1, t)
        y2 = self.conv_bn(prefix + ".cv2.1", t, cat.slice(c3, c4), 3)
        t2 = pb.alloc(x.B, x.H, x.W, c4)
        self.repncsp(prefix + ".cv3.0", y2, t2)
        self.conv_bn(prefix + ".cv3.1", t2, cat.slice(c3 + c4, c4), 3)
        if out is None:
            out = pb.alloc(x.B, x.H, x.W, c2)
        return self.conv_bn(prefix + ".cv4", cat, out, 1, res=res)

    def adown(self, prefix, x: View, out: View = None, res: View = None) -> View:
        pb = self.pb
        c = self.cout(prefix + ".cv1")
        half = x.C // 2
        ap = pb.alloc(x.B, x.H - 1, x.W - 1, x.C)
        pb.avgpool2(x, ap)
        Ho, Wo = (x.H - 2) // 2 + 1, (x.W - 2) // 2 + 1
        if out is None:
            out = pb.alloc(x.B, Ho, Wo, 2 * c)
        self.conv_bn(prefix + ".cv1", ap.slice(0, half), out.slice(0, c), 3, 2,
                     res=res.slice(0, c) if res is not None else None)
        mp = pb.alloc(x.B, Ho, Wo, half)
        pb.maxpool(ap.slice(half, half), mp, 3, 2, 1)
        self.conv_bn(prefix + ".cv2", mp, out.slice(c, c), 1, 1,
                     res=res.slice(c, c) if res is not None else None)
        return out

    def sppelan(self, prefix, x: View, out: View) -> View:
        pb = self.pb
        c3 = self.cout(prefix + ".cv1")
        cat = pb.alloc(x.B, x.H, x.W, 4 * c3)
        self.conv_bn(prefix + ".cv1", x, cat.slice(0, c3), 1)
        # three cascaded 5x5 pools.  One launch computing all three as 5 / 9 / 13-tap windows (exact for max) was built and measured in
        # round 5: 169 taps per output instead of 75 cost more than the two launches it saved (pools of a batch-1 plan 78.8 vs 59.1 us,
        # batch 8 190.7 vs 159.0, profiles/r5_s2_detector_per_op_*.txt) — removed
        for i in range(3):
            pb.maxpool(cat.slice(i * c3, c3), cat.slice((i + 1) * c3, c3), 5, 1, 2)
        return self.conv_bn(prefix + ".cv5", cat, out, 1)

    def cblinear(self, prefix, x: View, splits: List[int]) -> List[View]:
        pb = self.pb
        assert self.cout(prefix) == sum(splits)
        buf = pb.alloc(x.B, x.H, x.W, sum(splits))
        self.conv(prefix, lambda: (self.sd[prefix + ".conv.weight"], self.sd[prefix + ".conv.bias"]), x, buf, 1,
                  act=L.ACT_NONE)
        views, off = [], 0
        for c in splits:
            views.append(buf.slice(off, c))
            off += c
        return views

    def cbfuse(self, srcs: List[View], H, W) -> View:
        """sum of nearest-resized routing tensors (the last CBFuse operand is added by the conv epilogue)."""
        pb = self.pb
        tmp = pb.alloc(srcs[0].B, H, W, srcs[0].C)
        # one launch (round 5; the accumulate-resize chain it replaces — bit-identical, tests/gpu_checks.py::check_pools — cost 17 launches
        # / 128.6 us per batch-1 detector pass against 7 / 48.0 us now, profiles/r5_s1_detector_per_op_b1_640.txt)
        return pb.resize_sum(srcs, tmp)

    # ------------------------------------------------------------ whole network
    def build(self, x: View):
        """x: [B, TH, TW, V] letterboxed input (channels >= 3 are zero).  Returns per-stride
        (cls View [*, nc], box View [*, 64])."""
        pb, sd = self.pb, self.sd
        B = x.B
        c64, c128 = self.cout("a1"), self.cout("a2")
        c256, c512, c1024 = self.cout("a3.cv4"), self.cout("a5.cv4"), self.cout("a7.cv4")
        H2, W2 = self.TH // 2, self.TW // 2
        H4, W4, H8, W8 = H2 // 2, W2 // 2, H2 // 4, W2 // 4
        H16, W16, H32, W32 = H2 // 8, W2 // 8, H2 // 16, W2 // 16

        # neck concat buffers are allocated first so producers can write into them
        catA = pb.alloc(B, H16, W16, self.cout("n29.cv5") + c1024)      # [up2(p5) | b25]
        catB = pb.alloc(B, H8, W8, self.cout("n32.cv4") + c512)         # [up2(p4) | b22]
        catC = pb.alloc(B, H16, W16, self.cout("n36.cv1") * 2 + self.cout("n32.cv4"))   # [ADown(p3) | p4]
        catD = pb.alloc(B, H32, W32, self.cout("n39.cv1") * 2 + self.cout("n29.cv5"))   # [ADown(n4) | p5]
        cp5, cp4 = self.cout("n29.cv5"), self.cout("n32.cv4")

        # auxiliary branch
        a1 = self.conv_bn("a1", x, None, 3, 2, cin_pad=x.C)
        a2 = self.conv_bn("a2", a1, None, 3, 2)
        a3 = self.elan("a3", a2)
        a5 = self.elan("a5", self.adown("a4", a3))
        a7 = self.elan("a7", self.adown("a6", a5))
        a9 = self.elan("a9", self.adown("a8", a7))
        r10 = self.cblinear("r10", a1, [c64])
        r11 = self.cblinear("r11", a3, [c64, c128])
        r12 = self.cblinear("r12", a5, [c64, c128, c256])
        r13 = self.cblinear("r13", a7, [c64, c128, c256, c512])
        r14 = self.cblinear("r14", a9, [c64, c128, c256, c512, c1024])

        # main branch (CBFuse sums enter as residual operands)
        f16 = self.cbfuse([r10[0], r11[0], r12[0], r13[0], r14[0]], H2, W2)
        b = self.conv_bn("b15", x, None, 3, 2, res=f16, cin_pad=x.C)
        f18 = self.cbfuse([r11[1], r12[1], r13[1], r14[1]], H4, W4)
        b = self.conv_bn("b17", b, None, 3, 2, res=f18)
        b = self.elan("b19", b)
        f21 = self.cbfuse([r12[2], r13[2], r14[2]], H8, W8)
        b = self.adown("b20", b, res=f21)
        b22 = self.elan("b22", b, out=catB.slice(cp4, c512))
        f24 = self.cbfuse([r13[3], r14[3]], H16, W16)
        b = self.adown("b23", b22, res=f24)
        b25 = self.elan("b25", b, out=catA.slice(cp5, c1024))
        f27 = self.cbfuse([r14[4]], H32, W32)
        b = self.adown("b26", b25, res=f27)
        b28 = self.elan("b28", b)

        # neck
        p5 = self.sppelan("n29", b28, catD.slice(catD.C - cp5, cp5))
        pb.resize_nearest(p5, catA.slice(0, cp5))
        p4 = self.elan("n32", catA, out=catC.slice(catC.C - cp4, cp4))
        pb.resize_nearest(p4, catB.slice(0, cp4))
        p3 = self.elan("n35", catB)
        self.adown("n36", p3, out=catC.slice(0, catC.C - cp4))
        n4 = self.elan("n38", catC)
        self.adown("n39", n4, out=catD.slice(0, catD.C - cp5))
        n5 = self.elan("n41", catD)

        # heads: merged 3x3 stems, block-diagonal grouped convs
        outs = []
        for i, f in enumerate((p3, n4, n5)):
            cb, cc = self.cout(f"head.cv2.{i}.0"), self.cout(f"head.cv3.{i}.0")
            stem = pb.alloc(B, f.H, f.W, cb + cc)

            def stem_w(i=i):
                wb, bb = self.fold(f"head.cv2.{i}.0")
                wc, bc = self.fold(f"head.cv3.{i}.0")
                return torch.cat([wb, wc], 0), torch.cat([bb, bc], 0)
            self.conv(f"head.stem.{i}", stem_w, f, stem, 3)

            # box branch (groups=4 convs expanded to block-diagonal dense)
            def box1(i=i, cb=cb):
                w, bias = self.fold(f"head.cv2.{i}.1")
                return self.dense_from_groups(w, cb // w.shape[1]), bias

            def box2(i=i, cb=cb):
                w = sd[f"head.cv2.{i}.2.weight"]
                return self.dense_from_groups(w, cb // w.shape[1]), sd[f"head.cv2.{i}.2.bias"]
            t = pb.alloc(B, f.H, f.W, cb)
            self.conv(f"head.cv2.{i}.1", box1, stem.slice(0, cb), t, 3)
            box = pb.alloc(B, f.H, f.W, sd[f"head.cv2.{i}.2.weight"].shape[0])
            self.conv(f"head.cv2.{i}.2", box2, t, box, 1, act=L.ACT_NONE)
            # class branch
            t2 = pb.alloc(B, f.H, f.W, cc)
            self.conv(f"head.cv3.{i}.1", lambda i=i: self.fold(f"head.cv3.{i}.1"), stem.slice(cb, cc), t2, 3)
            cls = pb.alloc(B, f.H, f.W, self.nc)
            self.conv(f"head.cv3.{i}.2", lambda i=i: (sd[f"head.cv3.{i}.2.weight"], sd[f"head.cv3.{i}.2.bias"]),
                      t2, cls, 1, act=L.ACT_NONE)
            outs.append((cls, box))
        self.debug = {"a1": a1, "a3": a3, "a9": a9, "b28": b28, "p5": p5, "p4": p4, "p3": p3, "n4": n4, "n5": n5}
        return outs
