"""CPU interpreter of omni_op_t lists (TEST INFRASTRUCTURE ONLY).

Executes the descriptors a PlanBuilder produced with plain torch ops on the CPU tensors they point
to, so the graph lowering (weight folding, channel-slice bookkeeping, CBFuse-as-residual, merged
GEMMs) can be checked against the oracle network without a GPU.  The HIP kernels themselves are
checked on the GPU (tests marked `gpu`).
"""
import torch
import torch.nn.functional as F

from omniparser_amd import _lib as L


def _tensor_map(keep):
    return {t.data_ptr(): t for t in keep}


def run_ops(ops, keep):
    tm = _tensor_map(keep)
    for op in ops:
        i, p = op.i, op.p
        if op.kind == L.OP_CONV:
            B, H, W, Cin, ldi, icoff, KH, KW, s, pad, Ho, Wo, Cout, ldo, ocoff, act, ldr, rcoff = [i[k] for k in range(18)]
            x = tm[p[0]].view(B, H, W, ldi)[..., icoff:icoff + Cin].permute(0, 3, 1, 2).float()
            w = tm[p[1]].view(Cout, KH, KW, Cin).permute(0, 3, 1, 2).float()
            b = tm[p[2]].float() if p[2] else None
            y = F.conv2d(x, w, b, stride=s, padding=pad)
            if op.f[0] != 0.0:
                y = y * op.f[0]
            if act == L.ACT_SILU:
                y = F.silu(y)
            elif act == L.ACT_GELU:
                y = F.gelu(y)
            if p[3]:
                y = y + tm[p[3]].view(B, Ho, Wo, ldr)[..., rcoff:rcoff + Cout].permute(0, 3, 1, 2).float()
            out = tm[p[4]].view(B, Ho, Wo, ldo)
            out[..., ocoff:ocoff + Cout] = y.permute(0, 2, 3, 1).to(out.dtype)
        elif op.kind in (L.OP_AVGPOOL2, L.OP_MAXPOOL, L.OP_RESIZE_NEAREST):
            B, H, W, C, ldi, icoff = [i[k] for k in range(6)]
            k, s, pad, Ho, Wo, ldo, ocoff, acc = i[6], i[8], i[9], i[10], i[11], i[13], i[14], i[18]
            x = tm[p[0]].view(B, H, W, ldi)[..., icoff:icoff + C].permute(0, 3, 1, 2).float()
            if op.kind == L.OP_AVGPOOL2:
                y = F.avg_pool2d(x, 2, 1, 0, False, True)
                Ho, Wo = H - 1, W - 1
            elif op.kind == L.OP_MAXPOOL:
                y = F.max_pool2d(x, k, s, pad)
            else:
                y = F.interpolate(x, size=(Ho, Wo), mode="nearest")
            out = tm[p[4]].view(B, Ho, Wo, ldo)
            y = y.permute(0, 2, 3, 1)
            if acc:
                y = out[..., ocoff:ocoff + C].float() + y
            out[..., ocoff:ocoff + C] = y.to(out.dtype)
        else:
            raise NotImplementedError(f"interp: op kind {op.kind}")
