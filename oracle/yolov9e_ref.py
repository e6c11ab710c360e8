"""ORACLE (test infrastructure, never shipped, never timed as the product).

Plain PyTorch fp32 restatement of the YOLOv9-E network the reference loads as an opaque
TorchScript blob (ref:util/yolov9.py:50,121; weights `icon_detect_v3/model.pt` are NOT in
/root/reference and not on this box).  Topology follows the public YOLOv9-E definition as
tabulated in SURVEY.md Appendix B; the output contract is the one ref:util/yolov9.py:92-96
consumes: [cls_s8, dist_s8, cls_s16, dist_s16, cls_s32, dist_s32] with cls=[B,nc,H/s,W/s]
logits and dist=[B,4,H/s,W/s] LTRB distances in stride units (DFL reduced inside).

PARITY UNPINNED for the network itself: the reference ships no weights, golden outputs or
tests for it.  Parameter/FLOP counts match the published 57.3 M / 189 GFLOP.
Weights are seeded-random and deliberately WELL CONDITIONED (build_random_detector): small BatchNorm gains keep the
net from amplifying rounding noise (CPU fp32 vs CPU fp64 logits differ by ~3e-5 of the logit spread; the round-1
stand-in was chaotic, 5e-3), and the score threshold is centred in a gap of the parity frames' anchor logits, so the
network-level parity tests assert fixed epsilons and identical boxes on EVERY frame.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def autopad(k, p=None):
    return k // 2 if p is None else p


class Conv(nn.Module):
    """Conv2d(bias=False) + BatchNorm2d(eps=1e-3) + SiLU."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p), groups=g, bias=False)
        self.bn = nn.BatchNorm2d(c2, eps=1e-3, momentum=0.03)
        self.act = nn.SiLU() if act else nn.Identity()

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))


class RepConvN(nn.Module):
    def __init__(self, c1, c2):
        super().__init__()
        self.conv1 = Conv(c1, c2, 3, 1, act=False)
        self.conv2 = Conv(c1, c2, 1, 1, act=False)
        self.act = nn.SiLU()

    def forward(self, x):
        return self.act(self.conv1(x) + self.conv2(x))


class RepNBottleneck(nn.Module):
    def __init__(self, c1, c2):
        super().__init__()
        self.cv1 = RepConvN(c1, c2)
        self.cv2 = Conv(c2, c2, 3, 1)

    def forward(self, x):
        return x + self.cv2(self.cv1(x))


class RepNCSP(nn.Module):
    def __init__(self, c1, c2, n=1):
        super().__init__()
        c_ = c2 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*(RepNBottleneck(c_, c_) for _ in range(n)))

    def forward(self, x):
        return self.cv3(torch.cat((self.m(self.cv1(x)), self.cv2(x)), 1))


class RepNCSPELAN4(nn.Module):
    def __init__(self, c1, c2, c3, c4, n=1):
        super().__init__()
        self.cv1 = Conv(c1, c3, 1, 1)
        self.cv2 = nn.Sequential(RepNCSP(c3 // 2, c4, n), Conv(c4, c4, 3, 1))
        self.cv3 = nn.Sequential(RepNCSP(c4, c4, n), Conv(c4, c4, 3, 1))
        self.cv4 = Conv(c3 + 2 * c4, c2, 1, 1)

    def forward(self, x):
        y = list(self.cv1(x).chunk(2, 1))
        y.extend(m(y[-1]) for m in (self.cv2, self.cv3))
        return self.cv4(torch.cat(y, 1))


class ADown(nn.Module):
    def __init__(self, c1, c2):
        super().__init__()
        self.c = c2 // 2
        self.cv1 = Conv(c1 // 2, self.c, 3, 2, 1)
        self.cv2 = Conv(c1 // 2, self.c, 1, 1, 0)

    def forward(self, x):
        x = F.avg_pool2d(x, 2, 1, 0, False, True)
        x1, x2 = x.chunk(2, 1)
        x1 = self.cv1(x1)
        x2 = F.max_pool2d(x2, 3, 2, 1)
        x2 = self.cv2(x2)
        return torch.cat((x1, x2), 1)


class SPPELAN(nn.Module):
    def __init__(self, c1, c2, c3):
        super().__init__()
        self.cv1 = Conv(c1, c3, 1, 1)
        self.cv5 = Conv(4 * c3, c2, 1, 1)

    def forward(self, x):
        y = [self.cv1(x)]
        for _ in range(3):
            y.append(F.max_pool2d(y[-1], 5, 1, 2))
        return self.cv5(torch.cat(y, 1))


class CBLinear(nn.Module):
    def __init__(self, c1, c2s):
        super().__init__()
        self.c2s = list(c2s)
        self.conv = nn.Conv2d(c1, sum(c2s), 1, 1, 0, bias=True)

    def forward(self, x):
        return self.conv(x).split(self.c2s, dim=1)


class CBFuse(nn.Module):
    def __init__(self, idx):
        super().__init__()
        self.idx = list(idx)

    def forward(self, xs):
        target = xs[-1].shape[2:]
        res = [F.interpolate(x[self.idx[i]], size=target, mode="nearest") for i, x in enumerate(xs[:-1])]
        return torch.sum(torch.stack(res + [xs[-1]]), dim=0)


class DetectHead(nn.Module):
    """Per-scale box (DFL, 16 bins) and class branches; returns [cls, dist] per scale."""

    reg_max = 16

    def __init__(self, nc, ch):
        super().__init__()
        self.nc = nc
        c2 = max(ch[0] // 4, self.reg_max * 4, 16)
        c2 = (c2 + 3) // 4 * 4
        c3 = max(ch[0], min(nc * 2, 128))
        self.cv2 = nn.ModuleList(
            nn.Sequential(Conv(x, c2, 3), Conv(c2, c2, 3, g=4), nn.Conv2d(c2, 4 * self.reg_max, 1, groups=4))
            for x in ch)
        self.cv3 = nn.ModuleList(
            nn.Sequential(Conv(x, c3, 3), Conv(c3, c3, 3), nn.Conv2d(c3, nc, 1)) for x in ch)
        self.register_buffer("proj", torch.arange(self.reg_max, dtype=torch.float32), persistent=False)

    def dfl(self, box):
        b, _, h, w = box.shape
        p = box.view(b, 4, self.reg_max, h, w).softmax(2)
        return (p * self.proj.view(1, 1, -1, 1, 1)).sum(2)

    def forward(self, feats):
        out = []
        for i, f in enumerate(feats):
            out.append(self.cv3[i](f))
            out.append(self.dfl(self.cv2[i](f)))
        return out


class YOLOv9E(nn.Module):
    """YOLOv9-E (inference graph).  `width` scales every channel count (1.0 = the real model)."""

    def __init__(self, nc=1, width=1.0):
        super().__init__()
        w = lambda c: max(8, int(round(c * width / 8)) * 8)
        c64, c128, c256, c512, c1024 = w(64), w(128), w(256), w(512), w(1024)
        E = RepNCSPELAN4
        self.nc = nc
        # auxiliary (reversible) branch
        self.a1 = Conv(3, c64, 3, 2)
        self.a2 = Conv(c64, c128, 3, 2)
        self.a3 = E(c128, c256, c128, c64, 2)
        self.a4 = ADown(c256, c256)
        self.a5 = E(c256, c512, c256, c128, 2)
        self.a6 = ADown(c512, c512)
        self.a7 = E(c512, c1024, c512, c256, 2)
        self.a8 = ADown(c1024, c1024)
        self.a9 = E(c1024, c1024, c512, c256, 2)
        # routing
        self.r10 = CBLinear(c64, [c64])
        self.r11 = CBLinear(c256, [c64, c128])
        self.r12 = CBLinear(c512, [c64, c128, c256])
        self.r13 = CBLinear(c1024, [c64, c128, c256, c512])
        self.r14 = CBLinear(c1024, [c64, c128, c256, c512, c1024])
        # main branch
        self.b15 = Conv(3, c64, 3, 2)
        self.f16 = CBFuse([0, 0, 0, 0, 0])
        self.b17 = Conv(c64, c128, 3, 2)
        self.f18 = CBFuse([1, 1, 1, 1])
        self.b19 = E(c128, c256, c128, c64, 2)
        self.b20 = ADown(c256, c256)
        self.f21 = CBFuse([2, 2, 2])
        self.b22 = E(c256, c512, c256, c128, 2)
        self.b23 = ADown(c512, c512)
        self.f24 = CBFuse([3, 3])
        self.b25 = E(c512, c1024, c512, c256, 2)
        self.b26 = ADown(c1024, c1024)
        self.f27 = CBFuse([4])
        self.b28 = E(c1024, c1024, c512, c256, 2)
        # neck
        self.n29 = SPPELAN(c1024, c512, c256)
        self.n32 = E(c512 + c1024, c512, c512, c256, 2)
        self.n35 = E(c512 + c512, c256, c256, c128, 2)
        self.n36 = ADown(c256, c256)
        self.n38 = E(c256 + c512, c512, c512, c256, 2)
        self.n39 = ADown(c512, c512)
        self.n41 = E(c512 + c512, c512, c1024, c512, 2)
        self.head = DetectHead(nc, (c256, c512, c512))

    def forward(self, x):
        a1 = self.a1(x)
        a3 = self.a3(self.a2(a1))
        a5 = self.a5(self.a4(a3))
        a7 = self.a7(self.a6(a5))
        a9 = self.a9(self.a8(a7))
        r10, r11, r12, r13, r14 = self.r10(a1), self.r11(a3), self.r12(a5), self.r13(a7), self.r14(a9)
        b = self.f16([r10, r11, r12, r13, r14, self.b15(x)])
        b = self.f18([r11, r12, r13, r14, self.b17(b)])
        b = self.b19(b)
        b = self.f21([r12, r13, r14, self.b20(b)])
        b22 = self.b22(b)
        b = self.f24([r13, r14, self.b23(b22)])
        b25 = self.b25(b)
        b = self.f27([r14, self.b26(b25)])
        b28 = self.b28(b)
        p5 = self.n29(b28)
        p4 = self.n32(torch.cat((F.interpolate(p5, scale_factor=2.0, mode="nearest"), b25), 1))
        p3 = self.n35(torch.cat((F.interpolate(p4, scale_factor=2.0, mode="nearest"), b22), 1))
        n4 = self.n38(torch.cat((self.n36(p3), p4), 1))
        n5 = self.n41(torch.cat((self.n39(n4), p5), 1))
        return self.head([p3, n4, n5])


def letterbox_tensor(img_u8, imgsz):
    """PIL LANCZOS letterbox exactly as ref:util/yolov9.py:73-87 -> [1,3,H,W] float32."""
    from PIL import Image
    from . import detector_ref as D
    x, _, _, _ = D.preprocess(Image.fromarray(img_u8), imgsz)
    return x


def _calibration_input(seeds=tuple(range(8)), noisy=0, noise_std=0.1, random_frames=0, native=0):
    """640x640 letterboxes of synthetic screenshots (+ optionally `noisy` of them with additive pixel noise and `random_frames`
    uniform-noise images: directions the clean frames never excite)."""
    from omniparser_amd.synth import synthetic_screenshot
    x = torch.cat([letterbox_tensor(synthetic_screenshot(sd), 640) for sd in seeds])
    g = torch.Generator().manual_seed(1234)
    extra = []
    if noisy:
        extra.append((x[:noisy] + noise_std * torch.randn(x[:noisy].shape, generator=g)).clamp(0, 1))
    if random_frames:
        extra.append(torch.rand((random_frames,) + tuple(x.shape[1:]), generator=g))
    for i in range(native):
        # 640x640 windows of the frames at NATIVE scale (the scale_img=True path runs the network on unscaled pixels)
        img = synthetic_screenshot(seeds[i % len(seeds)])
        y0, x0 = (137 * i) % (img.shape[0] - 640), (411 * i) % (img.shape[1] - 640)
        extra.append(torch.from_numpy(img[y0:y0 + 640, x0:x0 + 640].copy()).permute(2, 0, 1)[None].float() / 255.0)
    return torch.cat([x] + extra)


def max_class_logits(model, x):
    res = []
    with torch.no_grad():
        for f in x.split(1):
            out = model(f)
            res.append(torch.cat([out[i].flatten(2) for i in (0, 2, 4)], 2).max(1).values.flatten())
    return torch.cat(res)


def calibrated_keys(state_dict):
    """state_dict entries that the calibration of build_random_detector computes (everything else is the seeded initialisation):
    BatchNorm running statistics and the last convolution of every class-head branch."""
    import re
    return [k for k in state_dict if k.endswith(("running_mean", "running_var", "num_batches_tracked")) or
            re.fullmatch(r"head\.cv3\.\d+\.2\.(weight|bias)", k)]


def build_random_detector(seed=0, nc=1, width=1.0, pass_rate=0.17, conf=0.05, margin_frames=(), calib_half=False, box_gain=1.0,
                          bn_gain=0.15, bn_shift=0.5, bn_shift_mean=1.0, calib_noisy=2, calib_random=1, calib_noise_std=0.1, calib_native=8,
                          calibration=None):
    """Seeded random YOLOv9-E that is WELL CONDITIONED, so that box-for-box parity can be asserted on every frame:

      * BatchNorm gains are small (gamma ~ 0.15) and shifts sizeable and POSITIVE (beta ~ 1.0 + 0.5 randn, "v5", round 5): every
        Conv+BN+SiLU then works on the upper, near-linear branch of SiLU (slope 0.8-1.0) for typical AND for outlier activations.
        Rounds 2-4 ("v4": gamma 0.25, beta ~ 0.5 randn, centred at 0) sat around x = 0 where the slope is ~0.5: content that is rare
        and strong — the binary "text" strips of the synthetic frames at NATIVE scale, which the 640x640 letterbox averages 3x3 but
        the scale_img=True path (1088x1920 network input) does not — then sees twice the per-layer gain of the typical signal the
        BatchNorm statistics were normalised on; the excess compounds with depth (kurtosis 20-80, 20-sigma activations, class-logit
        spread 5-23 instead of 1.5) and the stand-in's OWN f32 and f64 evaluations disagreed by 1e-2 ... 8e-2 in the heads at
        1088x1920.  A 3x3 box blur of the input removed the effect entirely (9.7e-6), adding native frames to the calibration batch
        or flooring the variances did not (.exp notes in DESIGN section 4).  With the shifted betas: 1.6e-5 ... 2.1e-5 at 1088x1920
        and 1.3e-5 ... 1.7e-5 at 640x640 on calibration and held-out seeds alike (the round-1 stand-in, gamma ~ 1: 5e-3);
      * running statistics are the POOLED statistics of one calibration batch: the eight synthetic bench screenshots (640x640
        letterboxes) plus — "v4", end of round 2 — two of them with additive pixel noise, one uniform-noise image and eight
        640x640 windows of the screenshots at native scale.  With the clean frames alone (v3) channels that are almost constant
        on flat GUI content got huge gains gamma/sigma; a frame that did excite them (3 of the 8 bench frames, every 640x480 frame,
        every native-resolution input) then carried activations of 20-60 sigma and amplified rounding noise to 1e-3 ... 1e-2 in the
        logits.  The extra frames put variance into those directions: measured f32-vs-f64 head difference 7e-6 ... 1.3e-5 on all
        eight bench frames and on the 640x480 frames at all three widths; 1088x1920 inputs remain the exception (3e-2 ... 7e-2:
        outliers near the image border, where native-scale windows do not reach);
      * the class head is rescaled to a logit spread of ~1.5 and biased so that ~`pass_rate` of the anchors exceed `conf`: ~1 200-1 450
        candidates, ~100-110 boxes per 1920x1080 screenshot;
      * `margin_frames` ([1,3,H,W] letterboxed inputs the parity tests run on): the threshold is centred in the widest gap
        between neighbouring DISTINCT anchor logits of the calibration + those frames around the requested pass rate, so no
        candidate sits within rounding distance of `conf` — the tests can then demand identical candidate sets
        unconditionally (model.margin / model.pass_rate record the half gap and the rate obtained)."""
    g = torch.Generator().manual_seed(seed)
    model = YOLOv9E(nc=nc, width=width)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, nn.Conv2d):
                fan_in = m.in_channels // m.groups * m.kernel_size[0] * m.kernel_size[1]
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * math.sqrt(2.0 / fan_in))
                if m.bias is not None:
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.copy_(bn_gain * (1.0 + 0.1 * torch.randn(m.weight.shape, generator=g)))
                m.bias.copy_(bn_shift_mean + bn_shift * torch.randn(m.bias.shape, generator=g))
        for m in model.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.momentum = 1.0
        model.train()
        for seq in model.head.cv2:   # icon-sized boxes: DFL mass centred near 1.5 strides per side, sizes that vary with the content
            bins = torch.arange(16, dtype=torch.float32)
            seq[-1].bias.copy_((-(bins - 1.5) ** 2 / 1.5).repeat(4))
            seq[-1].weight.mul_(box_gain)
        if calibration is not None:
            # `calibration` = the tensors a previous run of the passes below produced (tools/standin_calibration/*.pt, a few MB):
            # the same blob, bit for bit, on every box and in seconds — the calibration itself costs ~45 full CPU forward passes and
            # its BatchNorm statistics depend on the host's reduction order in the last bits
            model.eval()
            for m in model.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.momentum = 0.03
            sd = model.state_dict()
            want = set(calibrated_keys(sd))
            have = {k for k in calibration if k not in ("margin", "pass_rate")}
            assert have == want, (sorted(want - have)[:5], sorted(have - want)[:5])
            for k in want:
                assert sd[k].shape == calibration[k].shape and sd[k].dtype == calibration[k].dtype, k
                sd[k].copy_(calibration[k])
            model.margin = float(calibration["margin"])
            model.pass_rate = float(calibration["pass_rate"])
            return model
        x = _calibration_input()
        xc = _calibration_input(noisy=calib_noisy, random_frames=calib_random, noise_std=calib_noise_std, native=calib_native) \
            if (calib_noisy or calib_random or calib_native) else x
        # running stats <- POOLED batch statistics of all eight frames (per-frame statistics would leave the frames' global
        # differences un-normalised: whole frames then sit above / below the score threshold).  The pass runs on the 2x
        # box-filtered frames: a quarter of the working set, same per-channel statistics to within a few percent.
        model(F.avg_pool2d(xc, 2) if calib_half else xc)
        model.eval()
        for m in model.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.momentum = 0.03
        # class head: logit spread 1.5, `pass_rate` quantile at logit(conf)
        logits = max_class_logits(model, x)
        gain = 1.5 / float(logits.std().clamp_min(1e-6))
        for seq in model.head.cv3:
            seq[-1].weight.mul_(gain)
            seq[-1].bias.mul_(gain)
        # threshold placement: the `pass_rate` quantile of the anchor logits of the calibration + parity frames, moved into the
        # WIDEST gap between neighbouring distinct logits nearby (+-3 % of the anchors), so that no candidate sits within
        # rounding distance of `conf`.  Flat regions (the letterbox padding alone is 44 % of a 640x640 input) produce big
        # clusters of IDENTICAL logits: a quantile that falls into such a cluster would let a 1e-6 perturbation flip
        # thousands of anchors, and a cluster must not end up above the threshold (it would flood NMS with junk boxes) —
        # gaps are therefore searched among distinct values, above the dominant cluster when it lies in the window.
        thr = math.log(conf / (1 - conf))
        frames = [x]
        if len(margin_frames):
            by_shape = {}
            for f in margin_frames:
                by_shape.setdefault(tuple(f.shape[1:]), []).append(f)
            frames += [torch.cat(fs) for fs in by_shape.values()]
        lgx = max_class_logits(model, x)                               # the 640x640 calibration frames define the pass rate
        lg = torch.cat([lgx] + [max_class_logits(model, f) for f in frames[1:]])
        v_lo = float(torch.quantile(lgx, max(1.0 - pass_rate - 0.03, 0.0)))
        v_hi = float(torch.quantile(lgx, min(1.0 - pass_rate + 0.03, 1.0)))
        xv, xc = torch.unique(lgx, return_counts=True)
        big = int(torch.argmax(xc))
        if xc[big] > 0.02 * lgx.numel() and v_lo <= float(xv[big]) <= v_hi:
            v_lo = float(xv[big])                                       # stay above the dominant cluster
        vals = torch.unique(lg)                                         # sorted distinct logits of ALL frames
        vals = vals[(vals >= v_lo) & (vals <= v_hi)]
        if vals.numel() < 2:
            vals = torch.tensor([v_lo, max(v_hi, v_lo + 1e-3)])
        gaps = vals[1:] - vals[:-1]
        k = int(torch.argmax(gaps))
        centre = 0.5 * float(vals[k] + vals[k + 1])
        for seq in model.head.cv3:
            seq[-1].bias.add_(thr - centre)
        model.margin = 0.5 * float(gaps[k])
        model.pass_rate = float((lgx > centre).float().mean())
    return model.eval()


def count_params(model):
    return sum(p.numel() for p in model.parameters())
