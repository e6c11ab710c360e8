"""Host-side plan construction: device buffers (torch tensors = plumbing) + omni_op_t descriptors.

Tensors are NHWC.  A `View` is a channel slice [coff, coff+C) of a buffer with `ld` channels per
pixel, so chunk/concat/split never copy: producers write into slices, consumers read slices.
"""
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib as L


def vec_width(dtype: int) -> int:
    return 4 if dtype == L.F32 else 8


def torch_dtype(dtype: int):
    return torch.float32 if dtype == L.F32 else torch.float16


@dataclass
class View:
    t: torch.Tensor          # [B, H, W, ld]
    coff: int
    C: int

    @property
    def B(self): return self.t.shape[0]
    @property
    def H(self): return self.t.shape[1]
    @property
    def W(self): return self.t.shape[2]
    @property
    def ld(self): return self.t.shape[3]
    @property
    def ptr(self): return self.t.data_ptr()

    def slice(self, off, c):
        assert 0 <= off and off + c <= self.C, (off, c, self.C)
        return View(self.t, self.coff + off, c)

    def torch(self):
        """Materialise as a torch NCHW float tensor (tests / debugging)."""
        return self.t[..., self.coff:self.coff + self.C].permute(0, 3, 1, 2).float()


class PlanBuilder:
    def __init__(self, device, dtype: int):
        self.device = torch.device(device)
        self.dtype = dtype
        self.V = vec_width(dtype)
        self.ops = []
        self.keep = []           # keep-alive for every tensor referenced by raw pointer
        # f32 plans run their long-K GEMMs on the split-f16 MFMA path (f32-class accuracy at the f16 matrix rate;
        # measured error vs f64 <= the exact-f32 MFMA path).  OMNI_CONV_SPLIT=0 selects v_mfma_f32_32x32x2_f32 everywhere.
        self.split = os.environ.get("OMNI_CONV_SPLIT", "1") == "1"
        self.split_weights = set()
        self.ws = None           # split-K workspace shared by all convs of the plan (ops run in order)
        self.ws_kib = 32 * 1024
        self.flops = 0           # 2*MAC of all conv ops (algorithmic work, for the roofline)
        self.bytes = 0           # algorithmic HBM bytes (each operand read once, output written once)

    # ---- memory
    def alloc(self, B, H, W, C, zero=False) -> View:
        fn = torch.zeros if zero else torch.empty
        t = fn((B, H, W, C), dtype=torch_dtype(self.dtype), device=self.device)
        self.keep.append(t)
        return View(t, 0, C)

    def raw(self, shape, dtype, zero=True) -> torch.Tensor:
        t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
        self.keep.append(t)
        return t

    def upload(self, host_tensor: torch.Tensor) -> torch.Tensor:
        t = host_tensor.contiguous().to(self.device)
        self.keep.append(t)
        return t

    @staticmethod
    def split_f16(w2d: torch.Tensor) -> torch.Tensor:
        """[Cout, K] f32 (K % 16 == 0) -> f16 [Cout, K/16, 32]: per 16-wide K block 16 hi halves then 16 lo halves,
        w = hi + lo * 2^-11 (conv_igemm.hip split-f16 path)."""
        cout, K = w2d.shape
        assert K % 16 == 0
        w2d = w2d.float().clamp(-65504.0, 65504.0)
        hi = w2d.to(torch.float16)
        lo = ((w2d - hi.float()) * 2048.0).to(torch.float16)
        return torch.cat([hi.view(cout, K // 16, 16), lo.view(cout, K // 16, 16)], -1).contiguous()

    def pack_weight(self, w: torch.Tensor, cin_pad: Optional[int] = None) -> torch.Tensor:
        """[Cout, Cin, kh, kw] f32 -> device [Cout, kh*kw*Cin'] in plan dtype (k = (r*kw+s)*Cin'+c)."""
        w = w.detach().float()
        cout, cin, kh, kw = w.shape
        if cin_pad and cin_pad > cin:
            w = torch.cat([w, w.new_zeros(cout, cin_pad - cin, kh, kw)], 1)
        w = w.permute(0, 2, 3, 1).reshape(cout, -1)
        cin_eff = max(cin, cin_pad or 0)
        if self.split and self.dtype == L.F32 and cin_eff % 32 == 0 and cin_eff * kh * kw >= 128:
            t = self.upload(self.split_f16(w))
            self.split_weights.add(t.data_ptr())
            return t
        return self.upload(w.to(torch_dtype(self.dtype)))

    # ---- ops
    def conv(self, x: View, w_packed: torch.Tensor, bias: Optional[torch.Tensor], out: View, k: int, s: int = 1,
             p: Optional[int] = None, act: int = L.ACT_NONE, res: Optional[View] = None, scale: float = 0.0):
        p = k // 2 if p is None else p
        Ho = (x.H + 2 * p - k) // s + 1
        Wo = (x.W + 2 * p - k) // s + 1
        assert (out.B, out.H, out.W) == (x.B, Ho, Wo), ((out.B, out.H, out.W), (x.B, Ho, Wo))
        cout = out.C
        is_split = w_packed.dtype == torch.float16 and self.dtype == L.F32
        if is_split:
            assert w_packed.numel() == 2 * cout * k * k * x.C and x.C % 32 == 0, (tuple(w_packed.shape), cout, k, x.C)
        else:
            assert tuple(w_packed.shape) == (cout, k * k * x.C), (tuple(w_packed.shape), cout, k, x.C)
        if res is not None:
            assert (res.B, res.H, res.W, res.C) == (out.B, out.H, out.W, out.C)
        b = None
        if bias is not None:
            b = bias if bias.device == self.device and bias.dtype == torch.float32 else self.upload(bias.detach().float())
            self.keep.append(b)
            assert b.numel() == cout
        if self.ws is None and self.device.type == "cuda":
            self.ws = self.raw((self.ws_kib * 256,), torch.float32, zero=False)
        op = L.make_op(
            L.OP_CONV, self.dtype,
            p=[x.ptr, w_packed.data_ptr(), b.data_ptr() if b is not None else None,
               res.ptr if res is not None else None, out.ptr, self.ws.data_ptr() if self.ws is not None else None],
            i={0: x.B, 1: x.H, 2: x.W, 3: x.C, 4: x.ld, 5: x.coff, 6: k, 7: k, 8: s, 9: p, 10: Ho, 11: Wo,
               12: cout, 13: out.ld, 14: out.coff, 15: act,
               16: res.ld if res is not None else 0, 17: res.coff if res is not None else 0,
               19: self.ws_kib if self.ws is not None else 0, 20: 1 if is_split else 0},
            f={0: scale})
        self.ops.append(op)
        self.keep.append(w_packed)
        M = x.B * Ho * Wo
        esz = 4 if self.dtype == L.F32 else 2
        self.flops += 2 * M * cout * k * k * x.C
        self.bytes += esz * (x.B * x.H * x.W * x.C + w_packed.numel() + M * cout * (2 if res is not None else 1))
        return out

    def _pool(self, kind, x: View, out: View, k=0, s=1, p=0, accumulate=0):
        assert x.C == out.C and x.B == out.B
        op = L.make_op(
            kind, self.dtype, p=[x.ptr, None, None, None, out.ptr],
            i={0: x.B, 1: x.H, 2: x.W, 3: x.C, 4: x.ld, 5: x.coff, 6: k, 8: s, 9: p, 10: out.H, 11: out.W,
               13: out.ld, 14: out.coff, 18: accumulate})
        self.ops.append(op)
        esz = 4 if self.dtype == L.F32 else 2
        self.bytes += esz * (x.B * x.H * x.W * x.C + out.B * out.H * out.W * out.C * (2 if accumulate else 1))
        return out

    def avgpool2(self, x: View, out: View):
        assert (out.H, out.W) == (x.H - 1, x.W - 1)
        return self._pool(L.OP_AVGPOOL2, x, out)

    def maxpool(self, x: View, out: View, k, s, p):
        assert (out.H, out.W) == ((x.H + 2 * p - k) // s + 1, (x.W + 2 * p - k) // s + 1)
        return self._pool(L.OP_MAXPOOL, x, out, k, s, p)

    def resize_nearest(self, x: View, out: View, accumulate=False):
        return self._pool(L.OP_RESIZE_NEAREST, x, out, accumulate=1 if accumulate else 0)

    def add_op(self, op):
        self.ops.append(op)

    def build(self) -> "L.Plan":
        plan = L.Plan(self.ops)
        plan._keep = self.keep     # tensors live as long as the plan
        return plan
