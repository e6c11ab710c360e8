"""Host-side plan construction: device buffers (torch tensors = plumbing) + omni_op_t descriptors.

Tensors are NHWC.  A `View` is a channel slice [coff, coff+C) of a buffer with `ld` channels per
pixel, so chunk/concat/split never copy: producers write into slices, consumers read slices.
"""
import math
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
    fmt: str = "f32"         # "f32" | "split": what the bytes hold right now (f32 plans only; set by the producing op).
                             # "split" = format B of csrc/gemm_dma.hip: per 16 channels 16 hi halves then 16 lo halves.

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
        return View(self.t, self.coff + off, c, self.fmt)

    def torch(self):
        """Materialise as a torch NCHW float tensor (tests / debugging)."""
        return self.t[..., self.coff:self.coff + self.C].permute(0, 3, 1, 2).float()


# Every device tensor a PlanBuilder creates, by base address: (weak reference, role).  role: "const" = uploaded content that must
# travel with an exported plan (weights, tables), "zero" = zero-initialised state, "scratch" = written before it is read.
# omniparser_amd/bundle.py resolves the raw pointers of a plan's ops against this registry when it exports a plan bundle for
# the model-level C entry points (include/omni_amd.h: omni_model_*).
import weakref

_TENSORS = {}


def _register(t: torch.Tensor, role: str) -> torch.Tensor:
    if not t.numel():
        return t
    nbytes = t.numel() * t.element_size()
    ent = _TENSORS.get(t.data_ptr())
    if ent is not None:
        old = ent[0]()
        # a LIVE entry at this base address that already covers the range stays (e.g. `t` is the first slice of a registered table:
        # overwriting the table's entry with the slice's shorter range would orphan every pointer into the rest of it)
        if old is not None and old.data_ptr() == t.data_ptr() and ent[2] >= nbytes and old is not t:
            return t
    if len(_TENSORS) > 4096 and len(_TENSORS) % 1024 == 0:
        live_tensors()                                     # purge entries of dead tensors now and then (the registry is process-global)
    _TENSORS[t.data_ptr()] = (weakref.ref(t), role, nbytes)
    return t


def live_tensors():
    """[(base address, nbytes, role, tensor)] of the registered tensors that are still alive, sorted by address."""
    out = []
    for ptr, (ref, role, nbytes) in list(_TENSORS.items()):
        t = ref()
        if t is None or t.data_ptr() != ptr:
            _TENSORS.pop(ptr, None)
            continue
        out.append((ptr, nbytes, role, t))
    return sorted(out, key=lambda r: r[0])


def conv_key(M, N, K, k, s) -> str:
    """key of a conv shape in a tuning table: rows x output channels x reduction length, kernel size, stride"""
    return f"{M}x{N}x{K}k{k}s{s}"


class PlanBuilder:
    workspace_on_host = False    # plans on CPU tensors get no split-K workspace (the plan interpreter needs none); the host emulation
                                 # of the kernels (tests/emu) sets it so that split-K launches are what the MI355X runs

    def __init__(self, device, dtype: int):
        self.device = torch.device(device)
        self.dtype = dtype
        self.V = vec_width(dtype)
        self.ops = []
        self.keep = []           # keep-alive for every tensor referenced by raw pointer
        # f32 plans run their long-K GEMMs on the split-f16 MFMA path (f32-class accuracy at the f16 matrix rate;
        # measured error vs f64 <= the exact-f32 MFMA path).  OMNI_CONV_SPLIT=0 selects v_mfma_f32_32x32x2_f32 everywhere.
        self.split = os.environ.get("OMNI_CONV_SPLIT", "1") == "1"
        self.ws = None           # split-K workspace shared by all convs of the plan (ops run in order)
        self.ws_kib = 32 * 1024
        # split-K arrival counters of the in-launch combine (OMNI_OP_CONV i24 / p6): zero-initialised, shared by all convs of the plan
        # like the workspace (a launch leaves them zero).  OPT-IN (OMNI_SPLITK_COMBINE=1): bit-identical to the reduce launch and 234
        # launches instead of 431 per detector pass, but measured SLOWER on the MI355X in both sessions that compared them
        # (profiles/r6_s10_splitk_combine_ab.txt: batch-1 pass 4.44 -> 4.83 ms, batch 8 +0.6 %, 1088x1920 +1 %, the 20 decode steps'
        # GEMMs 17.2 -> 18.3 ms): the last arriver's serial ticket + write-through read-back costs more than a 5 us reduce launch.
        self.cnt = None
        self.n_cnt = 4096 if os.environ.get("OMNI_SPLITK_COMBINE", "0") == "1" else 0
        self.conv_tuning = None  # {conv_key: (tile code, split-K count)} for the split-f16 conv kernel, or None = the launcher's heuristic
        self.reuse = False       # lifetime reuse of released scratch tensors (see `release`)
        self._free, self._released, self._pins = [], set(), []
        self.reused_bytes = 0    # bytes handed out from released tensors instead of fresh allocations
        self.alloc_log = []      # every tensor `alloc` / `raw` handed out, in call order (what `arena` replays)
        self.arena = None        # iterator over ANOTHER builder's `alloc_log`: this builder allocates nothing, its k-th tensor is a
                                 # prefix view of that builder's k-th tensor (see `_from_arena`)
        self.flops = 0           # 2*MAC of all conv ops (algorithmic work, for the roofline)
        self.bytes = 0           # algorithmic HBM bytes (each operand read once, output written once)

    # ---- memory
    def alloc(self, B, H, W, C, zero=False) -> View:
        tdt = torch_dtype(self.dtype)
        if self.arena is not None:
            return View(self._from_arena((B, H, W, C), tdt), 0, C)
        if not zero:
            t = self._from_free((B, H, W, C), tdt)
            if t is not None:
                self.alloc_log.append(t)
                return View(t, 0, C)
        fn = torch.zeros if zero else torch.empty
        t = fn((B, H, W, C), dtype=tdt, device=self.device)
        self.keep.append(_register(t, "zero" if zero else "scratch"))
        self.alloc_log.append(t)
        return View(t, 0, C)

    def raw(self, shape, dtype, zero=True) -> torch.Tensor:
        if self.arena is not None:
            return self._from_arena(tuple(shape), dtype)
        if not zero:
            t = self._from_free(tuple(shape), dtype)
            if t is not None:
                self.alloc_log.append(t)
                return t
        t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
        self.keep.append(_register(t, "zero" if zero else "scratch"))
        self.alloc_log.append(t)
        return t

    # A plan over FEWER ROWS in the buffers of an existing plan (florence.py::_CaptionPlans.encode_rows: the last micro-batch of a
    # caption batch runs an encode graph of exactly its row count instead of a padded bucket).  The two builders make the same calls
    # in the same order — the op list of a network does not depend on the batch size — so tensor k of this builder is the same logical
    # tensor as tensor k of the arena's builder with a smaller leading dimension: it takes the FIRST bytes of that tensor.  Lifetimes
    # are those of the arena's plan (two of its tensors overlap only where lifetime reuse let them, and then the same two logical
    # tensors overlap here), zero-initialised tensors keep the arena's content (both plans write the same rows of them), nothing is
    # allocated, registered or released.  The arena's plan and this one must not run concurrently: they are issued on one stream.
    def _from_arena(self, shape, dtype):
        src = next(self.arena, None)
        if src is None:
            raise RuntimeError("arena replay: more allocations than the plan that owns the buffers made")
        need = math.prod(shape) * torch.empty((), dtype=dtype).element_size()
        have = src.numel() * src.element_size()
        if need > have or not src.is_contiguous():
            raise RuntimeError(f"arena replay: tensor of shape {tuple(shape)} ({need} B) does not fit the arena's {tuple(src.shape)} ({have} B)")
        t = src.view(-1).view(torch.uint8)[:need].view(dtype).view(shape)
        self.alloc_log.append(t)
        return t

    # Lifetime reuse of scratch tensors (`reuse = True`; off unless the plan's author turns it on).  Ops of a plan run in program
    # order on ONE stream, so a scratch tensor whose last reader has been added to the plan may back any tensor allocated later:
    # `release` returns its bytes to a free list, `alloc` / `raw` (zero=False) carve the best-fitting free block (256-byte granules,
    # the rest of the block stays free).  Only whole registered tensors ever enter the list, so the registry above (plan export)
    # keeps seeing each address range once; a carved tensor is a view of the block it came from.
    def release(self, *tensors):
        if not self.reuse or self.arena is not None:
            return
        for t in tensors:
            t = t.t if isinstance(t, View) else t
            if t is None or id(t) in self._released or t.numel() == 0:
                continue
            self._released.add(id(t))
            self._pins.append(t)                       # keeps id(t) unique for the lifetime of the builder
            nbytes = t.numel() * t.element_size() // 256 * 256
            if nbytes:
                self._free.append((t.view(-1).view(torch.uint8)[:nbytes], nbytes))

    def _from_free(self, shape, dtype):
        if not self.reuse or not self._free:
            return None
        need = math.prod(shape) * torch.empty((), dtype=dtype).element_size()
        cand = [(nb, k) for k, (_, nb) in enumerate(self._free) if nb >= need]
        if not cand or need == 0:
            return None
        nb, k = min(cand)
        block, _ = self._free.pop(k)
        rest = (nb - need) // 256 * 256
        if rest:
            self._free.append((block[nb - rest:], rest))
        self.reused_bytes += need
        return block[:need].view(dtype).view(shape)

    def upload(self, host_tensor: torch.Tensor) -> torch.Tensor:
        t = host_tensor.contiguous().to(self.device)
        self.keep.append(_register(t, "const"))
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

    @staticmethod
    def split_f16_b(w2d: torch.Tensor):
        """[Cout, K] f32 -> (f16 [Cout, K/16, 32], 2^-k): "format B" of csrc/gemm_dma.hip.  W' = W * 2^k with the largest
        |W'| in [2^12, 2^13) so that the lo parts of all but vanishing weights are normal f16 numbers; per 16-wide K block
        16 hi halves then 16 lo halves with W' = hi + lo (lo NOT rescaled: one accumulator takes all three products)."""
        cout, K = w2d.shape
        assert K % 16 == 0
        w2d = w2d.double()
        amax = float(w2d.abs().max())
        k = 0 if amax == 0.0 else 12 - math.floor(math.log2(amax))
        k = max(-24, min(24, k))
        ws = (w2d * (2.0 ** k)).float()
        hi = ws.to(torch.float16)
        lo = (ws - hi.float()).to(torch.float16)
        return torch.cat([hi.view(cout, K // 16, 16), lo.view(cout, K // 16, 16)], -1).contiguous(), 2.0 ** -k

    KPERM16 = (0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15)

    def pack_weight_dma(self, w: torch.Tensor, kperm: bool = False) -> torch.Tensor:
        """[Cout, Cin] or [Cout, Cin, 1, 1] f32 -> device format-B weight for the pre-split LDS-DMA GEMM (f32 plans).
        kperm: the K axis is permuted inside every 16-group to the order in which a 32x32 MFMA accumulator tile holds it as the
        column operand of the next product (fc2 of OMNI_OP_MLP_FUSED: position 8h + j holds channel 8 (j // 4) + 4h + j % 4)."""
        w2d = w.detach().float().reshape(w.shape[0], -1)
        if kperm:
            cout, K = w2d.shape
            assert K % 16 == 0
            w2d = w2d.view(cout, K // 16, 16)[:, :, list(self.KPERM16)].reshape(cout, K).contiguous()
        t, oscale = self.split_f16_b(w2d)
        t = self.upload(t)
        t.omni_fmt, t.omni_oscale = 2, oscale
        return t

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
            t.omni_fmt = 1
            return t
        return self.upload(w.to(torch_dtype(self.dtype)))

    # ---- ops
    def conv(self, x: View, w_packed: torch.Tensor, bias: Optional[torch.Tensor], out: View, k: int, s: int = 1,
             p: Optional[int] = None, act: int = L.ACT_NONE, res: Optional[View] = None, scale: float = 0.0,
             out_split: bool = False):
        p = k // 2 if p is None else p
        Ho = (x.H + 2 * p - k) // s + 1
        Wo = (x.W + 2 * p - k) // s + 1
        assert (out.B, out.H, out.W) == (x.B, Ho, Wo), ((out.B, out.H, out.W), (x.B, Ho, Wo))
        cout = out.C
        wfmt = getattr(w_packed, "omni_fmt", 0) if self.dtype == L.F32 else 0     # 0 plain, 1 split (in-loop), 2 format B (LDS-DMA GEMM)
        if wfmt:
            assert w_packed.dtype == torch.float16 and w_packed.numel() == 2 * cout * k * k * x.C and x.C % 32 == 0, \
                (tuple(w_packed.shape), cout, k, x.C)
        else:
            assert tuple(w_packed.shape) == (cout, k * k * x.C), (tuple(w_packed.shape), cout, k, x.C)
        if wfmt == 2:
            assert k == 1 and s == 1 and p == 0 and scale == 0.0 and cout % 128 == 0, "format-B weights: pointwise layers, Cout % 128 == 0"
            assert x.fmt == "split", "the LDS-DMA GEMM reads a pre-split input: its producer must emit format B (or insert split_convert)"
            assert x.ld % 16 == 0 and x.coff % 16 == 0 and not (out_split and (res is not None or out.ld % 16 or out.coff % 16))
        else:
            assert x.fmt == "f32" and not out_split, "register-staged conv kernels read and write f32"
        if res is not None:
            assert (res.B, res.H, res.W, res.C) == (out.B, out.H, out.W, out.C) and res.fmt == "f32"
        b = None
        if bias is not None:
            b = bias if bias.device == self.device and bias.dtype == torch.float32 else self.upload(bias.detach().float())
            ent = _TENSORS.get(b.data_ptr())
            if ent is None or ent[0]() is not b:  # (an entry of a DEAD tensor that lived at this address does not count)
                _register(b, "const")            # a caller-owned device tensor: exported with the plan like an uploaded one
            self.keep.append(b)
            assert b.numel() == cout
        if self.ws is None and (self.device.type == "cuda" or self.workspace_on_host):
            self.ws = self.raw((self.ws_kib * 256,), torch.float32, zero=False)
        if self.cnt is None and self.ws is not None and self.n_cnt and wfmt == 1:
            self.cnt = self.raw((self.n_cnt,), torch.int32, zero=True)
        cnt = self.cnt if (wfmt == 1 and self.cnt is not None) else None
        # tile / split-K override of the split-f16 conv kernel from the plan's tuning table (None: the launcher's heuristic)
        tune = (0, 0)
        if wfmt == 1 and self.conv_tuning:
            tune = tuple(self.conv_tuning.get(conv_key(x.B * Ho * Wo, cout, k * k * x.C, k, s), (0, 0)))
        op = L.make_op(
            L.OP_CONV, self.dtype,
            p=[x.ptr, w_packed.data_ptr(), b.data_ptr() if b is not None else None,
               res.ptr if res is not None else None, out.ptr, self.ws.data_ptr() if self.ws is not None else None,
               cnt.data_ptr() if cnt is not None else None],
            i={0: x.B, 1: x.H, 2: x.W, 3: x.C, 4: x.ld, 5: x.coff, 6: k, 7: k, 8: s, 9: p, 10: Ho, 11: Wo,
               12: cout, 13: out.ld, 14: out.coff, 15: act,
               16: res.ld if res is not None else 0, 17: res.coff if res is not None else 0,
               19: self.ws_kib if self.ws is not None else 0, 20: wfmt, 21: 1 if out_split else 0,
               22: tune[0], 23: tune[1], 24: cnt.numel() if cnt is not None else 0},
            f={0: scale, 1: getattr(w_packed, "omni_oscale", 0.0) if wfmt == 2 else 0.0})
        self.ops.append(op)
        self.keep.append(w_packed)
        out.fmt = "split" if out_split else "f32"
        M = x.B * Ho * Wo
        esz = 4 if self.dtype == L.F32 else 2
        self.flops += 2 * M * cout * k * k * x.C
        self.bytes += esz * (x.B * x.H * x.W * x.C + w_packed.numel() // (2 if wfmt else 1) + M * cout * (2 if res is not None else 1))
        return out

    def pack_weight_patch(self, w: torch.Tensor, ld: int = 4) -> torch.Tensor:
        """[Cout, Cin <= ld, k, k] f32 (k <= 8) -> split-f16 weight of the ROW-PATCH form of the convolution (conv_patch):
        [Cout][k rows][8 pixels][ld channels], zero beyond tap k - 1 and beyond channel Cin - 1."""
        w = w.detach().float()
        cout, cin, kh, kw = w.shape
        assert kh == kw and kh <= 8 and cin <= ld == 4
        full = w.new_zeros(cout, kh, 8, ld)
        full[:, :, :kw, :cin] = w.permute(0, 2, 3, 1)
        t = self.upload(self.split_f16(full.reshape(cout, kh * 8 * ld)))
        t.omni_fmt = 1
        return t

    def conv_patch(self, x: View, w_patch: torch.Tensor, bias: torch.Tensor, out: View, k: int, s: int, p: int, act: int = L.ACT_NONE):
        """k x k convolution (stride s, padding p) over the ld = 4 stored channels of x on the split-f16 MFMA kernel, as a k x 1
        convolution over 8 consecutive pixels (OMNI_OP_CONV i25 = 1; csrc/conv_igemm.hip "row-patch mode"): K = 32 k on the f16
        matrix pipe instead of 4 k^2 on the exact-f32 one."""
        assert self.dtype == L.F32 and x.fmt == "f32" and x.ld == 4 and x.coff == 0 and x.C == 4 and k <= 8
        Ho = (x.H + 2 * p - k) // s + 1
        Wo = (x.W + 2 * p - k) // s + 1
        cout = out.C
        assert (out.B, out.H, out.W) == (x.B, Ho, Wo) and w_patch.numel() == 2 * cout * k * 32 and getattr(w_patch, "omni_fmt", 0) == 1
        assert bias.device.type == self.device.type and bias.dtype == torch.float32 and bias.numel() == cout
        if self.ws is None and (self.device.type == "cuda" or self.workspace_on_host):
            self.ws = self.raw((self.ws_kib * 256,), torch.float32, zero=False)
        self.ops.append(L.make_op(
            L.OP_CONV, self.dtype,
            p=[x.ptr, w_patch.data_ptr(), bias.data_ptr(), None, out.ptr, self.ws.data_ptr() if self.ws is not None else None],
            i={0: x.B, 1: x.H, 2: x.W, 3: 32, 4: 4, 5: 0, 6: k, 7: 1, 8: s, 9: p, 10: Ho, 11: Wo, 12: cout, 13: out.ld, 14: out.coff,
               15: act, 19: self.ws_kib if self.ws is not None else 0, 20: 1, 25: 1}))
        self.keep += [w_patch, bias]
        out.fmt = "f32"
        M = x.B * Ho * Wo
        self.flops += 2 * M * cout * k * k * x.C           # ALGORITHMIC work: the k x k x 4 taps, not the zero-padded 32 k
        self.bytes += 4 * (x.B * x.H * x.W * x.C + cout * k * k * x.C + M * cout)
        return out

    def mlp_fused(self, x: View, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, res: View, out: View):
        """out = res + fc2(GELU(fc1(x))) as ONE op (csrc/gemm_dma.hip::mlp_fused_kernel): x in format B, w1 / w2 from
        pack_weight_dma (w2 with kperm=True), biases f32 on the device; hidden activations never reach HBM."""
        C, hid = x.C, b1.numel()
        assert self.dtype == L.F32 and x.fmt == "split" and res.fmt == "f32" and (C, hid) == (128, 512), (x.fmt, res.fmt, C, hid)
        assert getattr(w1, "omni_fmt", 0) == 2 and getattr(w2, "omni_fmt", 0) == 2 and w1.numel() == 2 * hid * C == w2.numel()
        assert (out.B, out.H, out.W, out.C) == (x.B, x.H, x.W, C) == (res.B, res.H, res.W, res.C) and b2.numel() == C
        rows = x.B * x.H * x.W
        self.ops.append(L.make_op(L.OP_MLP_FUSED, self.dtype,
                                  p=[x.ptr, w1.data_ptr(), b1.data_ptr(), res.ptr, out.ptr, w2.data_ptr(), b2.data_ptr()],
                                  i={0: rows, 1: 1, 3: C, 4: x.ld, 5: x.coff, 12: hid, 13: out.ld, 14: out.coff, 16: res.ld, 17: res.coff},
                                  f={1: w1.omni_oscale, 2: w2.omni_oscale}))
        self.keep += [w1, w2, b1, b2]
        out.fmt = "f32"
        self.flops += 2 * rows * C * hid * 2
        self.bytes += 4 * (rows * C * 3 + 2 * C * hid)          # h in, residual in, y out, both weight matrices
        return out

    def split_convert(self, x: View, out: Optional[View] = None) -> View:
        """f32 channel slice -> format B (in place by default): for GEMM inputs whose producer does not emit the split format."""
        out = out or x
        assert self.dtype == L.F32 and x.fmt == "f32" and x.C % 16 == 0 and (out.B, out.H, out.W, out.C) == (x.B, x.H, x.W, x.C)
        self.ops.append(L.make_op(L.OP_SPLIT_CONVERT, self.dtype, p=[x.ptr, None, None, None, out.ptr],
                                  i={0: x.B * x.H * x.W, 1: 1, 3: x.C, 4: x.ld, 5: x.coff, 13: out.ld, 14: out.coff}))
        self.bytes += 8 * x.B * x.H * x.W * x.C
        out.fmt = "split"
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

    def resize_sum(self, srcs, out: View):
        """out = ((resize(s0) + resize(s1)) + ...) — CBFuse as ONE launch (OMNI_OP_RESIZE_NEAREST with i17 sources, at most 5): the same
        partial sums in the same order as a chain of accumulate-resize launches, one pass over the output instead of len(srcs)."""
        assert 1 <= len(srcs) <= 5 and all(s.C == out.C and s.B == out.B for s in srcs)
        if len(srcs) == 1:
            return self.resize_nearest(srcs[0], out)
        x = srcs[0]
        slots = ((19, 20, 21, 22), (23, 24, 25, 26), (27, 28, 29, 30), (7, 12, 15, 16))
        pidx = (1, 2, 3, 5)
        p = [x.ptr, None, None, None, out.ptr, None]
        i = {0: x.B, 1: x.H, 2: x.W, 3: x.C, 4: x.ld, 5: x.coff, 10: out.H, 11: out.W, 13: out.ld, 14: out.coff, 17: len(srcs)}
        for k, sv in enumerate(srcs[1:]):
            p[pidx[k]] = sv.ptr
            i.update({slots[k][0]: sv.H, slots[k][1]: sv.W, slots[k][2]: sv.ld, slots[k][3]: sv.coff})
        self.ops.append(L.make_op(L.OP_RESIZE_NEAREST, self.dtype, p=p, i=i))
        esz = 4 if self.dtype == L.F32 else 2
        self.bytes += esz * (sum(s.B * s.H * s.W * s.C for s in srcs) + out.B * out.H * out.W * out.C)
        return out

    def add_op(self, op):
        self.ops.append(op)

    def build(self) -> "L.Plan":
        plan = L.Plan(self.ops)
        plan._keep = self.keep     # tensors live as long as the plan
        return plan
