"""CPU interpreter of omni_op_t lists (TEST INFRASTRUCTURE ONLY).

Executes the descriptors a PlanBuilder produced with plain torch / numpy ops on the CPU tensors they
point to.  Two uses: (1) check the graph lowering (weight folding, slice bookkeeping, merged GEMMs,
decode-loop wiring) against the oracle models without a GPU; (2) per-op reference for the `gpu` tests.
Semantics are written independently of the HIP kernels, from the reference/hf code each op replaces.
"""
import bisect

import numpy as np
import torch
import torch.nn.functional as F

from omniparser_amd import _lib as L


class Mem:
    """raw address -> (tensor, element offset)."""

    def __init__(self, keep):
        uniq = {}
        for t in keep:
            if isinstance(t, torch.Tensor) and t.numel():
                uniq[t.data_ptr()] = t
        self.starts = sorted(uniq)
        self.tensors = [uniq[s] for s in self.starts]

    def at(self, ptr, dtype=None):
        """flat 1-D view starting at ptr (dtype defaults to the owning tensor's)."""
        if not ptr:
            return None
        i = bisect.bisect_right(self.starts, ptr) - 1
        t = self.tensors[i]
        off = ptr - self.starts[i]
        assert 0 <= off < t.numel() * t.element_size(), "pointer outside every known tensor"
        flat = t.reshape(-1).view(torch.uint8)[off:]
        dt = dtype or t.dtype
        n = flat.numel() // torch.empty((), dtype=dt).element_size()
        return flat[: n * torch.empty((), dtype=dt).element_size()].view(dt)



# ---- "format B" (csrc/gemm_dma.hip): per 16 channels 64 bytes = 16 hi halves then 16 lo halves, value = hi + lo
def _rtz_f16(x: torch.Tensor) -> torch.Tensor:
    """f32 -> f16 rounding toward zero (v_cvt_pkrtz_f16_f32), input already clamped to the f16 range."""
    h = x.to(torch.float16)
    bits = h.view(torch.int16).clone()
    over = h.float().abs() > x.abs()
    bits[over] -= 1                      # one ulp toward zero (sign-magnitude encoding)
    return bits.view(torch.float16)


def split_encode(x: torch.Tensor) -> torch.Tensor:
    """[rows, C] f32 (C % 16 == 0) -> [rows, C] f32-typed buffer holding format B."""
    rows, C = x.shape
    x = x.float().clamp(-65504.0, 65504.0)
    hi = _rtz_f16(x)
    lo = _rtz_f16(x - hi.float())
    out = torch.cat([hi.view(rows, C // 16, 16), lo.view(rows, C // 16, 16)], -1).contiguous()
    return out.view(rows, 2 * C).view(torch.float32)


def split_decode(buf: torch.Tensor) -> torch.Tensor:
    """inverse view: [rows, C] f32-typed buffer in format B -> f32 values."""
    rows, C = buf.shape
    hl = buf.contiguous().view(torch.float16).view(rows, C // 16, 2, 16).float()
    return (hl[:, :, 0] + hl[:, :, 1]).reshape(rows, C)


KPERM16 = (0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15)     # OMNI_OP_MLP_FUSED: K order of w2 inside a 16-group (include/omni_amd.h)


def _tdt(op):
    return torch.float32 if op.dtype == L.F32 else torch.float16


def run_ops(ops, keep):
    m = Mem(keep)
    for op in ops:
        run_op(op, m)


def run_op(op, m):
    i, p, f = op.i, op.p, op.f
    dt = _tdt(op)
    k = op.kind
    if k == L.OP_CONV and i[20] == 2:
        # pre-split LDS-DMA GEMM: x and w in format B, optional format-B output
        B, H, W, Cin, ldi, icoff = [i[j] for j in range(6)]
        Cout, ldo, ocoff, act, ldr, rcoff = [i[j] for j in range(12, 18)]
        M = B * H * W
        x = split_decode(m.at(p[0], torch.float32)[: M * ldi].view(M, ldi)[:, icoff:icoff + Cin])
        w = split_decode(m.at(p[1], torch.float16)[: 2 * Cout * Cin].view(torch.float32).view(Cout, Cin)) * f[1]
        y = x @ w.t()
        if p[2]:
            y = y + m.at(p[2], torch.float32)[:Cout]
        if act == L.ACT_GELU:
            y = F.gelu(y)
        elif act == L.ACT_SILU:
            y = F.silu(y)
        if p[3]:
            y = y + m.at(p[3], torch.float32)[: M * ldr].view(M, ldr)[:, rcoff:rcoff + Cout]
        out = m.at(p[4], torch.float32)[: M * ldo].view(M, ldo)
        out[:, ocoff:ocoff + Cout] = split_encode(y) if i[21] else y
    elif k == L.OP_MLP_FUSED:
        rows, C, ldi, icoff, hid, ldo, ocoff, ldr, rcoff = i[0] * max(i[1], 1), i[3], i[4], i[5], i[12], i[13], i[14], i[16], i[17]
        x = split_decode(m.at(p[0], torch.float32)[: rows * ldi].view(rows, ldi)[:, icoff:icoff + C])
        w1 = split_decode(m.at(p[1], torch.float16)[: 2 * hid * C].view(torch.float32).view(hid, C)) * f[1]
        w2p = split_decode(m.at(p[5], torch.float16)[: 2 * hid * C].view(torch.float32).view(C, hid)) * f[2]
        inv = [KPERM16.index(j) for j in range(16)]                     # stored position of original channel j inside its 16-group
        w2 = w2p.view(C, hid // 16, 16)[:, :, inv].reshape(C, hid)
        h = F.gelu(x @ w1.t() + m.at(p[2], torch.float32)[:hid])
        h = split_decode(split_encode(h))                               # the kernel hands fc2 the (hi, lo) halves of the activations
        y = h @ w2.t() + m.at(p[6], torch.float32)[:C] + m.at(p[3], torch.float32)[: rows * ldr].view(rows, ldr)[:, rcoff:rcoff + C]
        m.at(p[4], torch.float32)[: rows * ldo].view(rows, ldo)[:, ocoff:ocoff + C] = y
    elif k == L.OP_SPLIT_CONVERT:
        rows, C, ldi, icoff, ldo, ocoff = i[0] * max(i[1], 1), i[3], i[4], i[5], i[13], i[14]
        x = m.at(p[0], torch.float32)[: rows * ldi].view(rows, ldi)[:, icoff:icoff + C].clone()
        m.at(p[4], torch.float32)[: rows * ldo].view(rows, ldo)[:, ocoff:ocoff + C] = split_encode(x)
    elif k == L.OP_CONV:
        B, H, W, Cin, ldi, icoff, KH, KW, s, pad, Ho, Wo, Cout, ldo, ocoff, act, ldr, rcoff = [i[j] for j in range(18)]
        if i[25]:
            # row-patch mode (include/omni_amd.h): a KH x KH convolution over the ldi = 4 stored channels, weights [Cout][KH][8 pixels][4]
            hl = m.at(p[1], torch.float16)[: 2 * Cout * KH * 32].view(Cout, KH * 2, 2, 16).float()
            w = (hl[:, :, 0] + hl[:, :, 1] / 2048.0).reshape(Cout, KH, 8, ldi)
            assert float(w[:, :, KH:].abs().max()) == 0.0
            w = w[:, :, :KH].permute(0, 3, 1, 2)
            x = m.at(p[0], dt)[: B * H * W * ldi].view(B, H, W, ldi).permute(0, 3, 1, 2).float()
            Cin, KW = ldi, KH
        else:
            x = m.at(p[0], dt)[: B * H * W * ldi].view(B, H, W, ldi)[..., icoff:icoff + Cin].permute(0, 3, 1, 2).float()
        if i[25]:
            pass
        elif i[20]:
            hl = m.at(p[1], torch.float16)[: 2 * Cout * KH * KW * Cin].view(Cout, KH * KW * Cin // 16, 2, 16).float()
            w = (hl[:, :, 0] + hl[:, :, 1] / 2048.0).reshape(Cout, KH, KW, Cin).permute(0, 3, 1, 2)
        else:
            w = m.at(p[1], dt)[: Cout * KH * KW * Cin].view(Cout, KH, KW, Cin).permute(0, 3, 1, 2).float()
        b = m.at(p[2], torch.float32)[:Cout] if p[2] else None
        y = F.conv2d(x, w, b, stride=s, padding=pad)
        if f[0] != 0.0:
            y = y * f[0]
        if act == L.ACT_SILU:
            y = F.silu(y)
        elif act == L.ACT_GELU:
            y = F.gelu(y)
        if p[3]:
            y = y + m.at(p[3], dt)[: B * Ho * Wo * ldr].view(B, Ho, Wo, ldr)[..., rcoff:rcoff + Cout].permute(0, 3, 1, 2).float()
        out = m.at(p[4], dt)[: B * Ho * Wo * ldo].view(B, Ho, Wo, ldo)
        out[..., ocoff:ocoff + Cout] = y.permute(0, 2, 3, 1).to(dt)
    elif k in (L.OP_AVGPOOL2, L.OP_MAXPOOL, L.OP_RESIZE_NEAREST):
        B, H, W, C, ldi, icoff = [i[j] for j in range(6)]
        kk, s, pad, Ho, Wo, ldo, ocoff, acc = i[6], i[8], i[9], i[10], i[11], i[13], i[14], i[18]
        x = m.at(p[0], dt)[: B * H * W * ldi].view(B, H, W, ldi)[..., icoff:icoff + C].permute(0, 3, 1, 2).float()
        if k == L.OP_AVGPOOL2:
            y = F.avg_pool2d(x, 2, 1, 0, False, True); Ho, Wo = H - 1, W - 1
        elif k == L.OP_MAXPOOL:
            y = F.max_pool2d(x, kk, s, pad)
        else:
            y = F.interpolate(x, size=(Ho, Wo), mode="nearest")
            if i[17] > 1:                       # CBFuse in one op: ((r(x0) + r(x1)) + ...) with every partial sum rounded to the plan's dtype
                slots = ((19, 20, 21, 22), (23, 24, 25, 26), (27, 28, 29, 30), (7, 12, 15, 16))
                for kx in range(i[17] - 1):
                    Hk, Wk, ldk, cok = (i[j] for j in slots[kx])
                    xk = m.at(p[(1, 2, 3, 5)[kx]], dt)[: B * Hk * Wk * ldk].view(B, Hk, Wk, ldk)[..., cok:cok + C].permute(0, 3, 1, 2).float()
                    y = y.to(dt).float() + F.interpolate(xk, size=(Ho, Wo), mode="nearest")
        out = m.at(p[4], dt)[: B * Ho * Wo * ldo].view(B, Ho, Wo, ldo)
        y = y.permute(0, 2, 3, 1)
        if acc:
            y = out[..., ocoff:ocoff + C].float() + y
        out[..., ocoff:ocoff + C] = y.to(dt)
    elif k == L.OP_DWCONV3:
        B, H, W, C = i[0], i[1], i[2], i[3]
        x = m.at(p[0], dt)[: B * H * W * C].view(B, H, W, C).permute(0, 3, 1, 2).float()
        w = m.at(p[1], dt)[: 9 * C].view(3, 3, C).permute(2, 0, 1).unsqueeze(1).float()
        b = m.at(p[2], torch.float32)[:C]
        y = F.conv2d(x, w, b, padding=1, groups=C) + x
        m.at(p[4], dt)[: B * H * W * C].view(B, H, W, C).copy_(y.permute(0, 2, 3, 1).to(dt))
    elif k == L.OP_DWCONV3_LN:
        B, H, W, C = i[0], i[1], i[2], i[3]
        x = m.at(p[0], dt)[: B * H * W * C].view(B, H, W, C).permute(0, 3, 1, 2).float()
        w = m.at(p[1], dt)[: 9 * C].view(3, 3, C).permute(2, 0, 1).unsqueeze(1).float()
        y1 = (F.conv2d(x, w, m.at(p[2], torch.float32)[:C], padding=1, groups=C) + x).permute(0, 2, 3, 1).to(dt)
        m.at(p[4], dt)[: B * H * W * C].view(B, H, W, C).copy_(y1)
        hn = F.layer_norm(y1.float(), (C,), m.at(p[5], torch.float32)[:C], m.at(p[6], torch.float32)[:C], f[0])
        m.at(p[3], dt)[: B * H * W * C].view(B, H, W, C).copy_(split_encode(hn.reshape(-1, C)).view(B, H, W, C) if i[6] else hn.to(dt))
    elif k == L.OP_LAYERNORM:
        rows, C, period = i[0] * max(i[1], 1), i[3], i[5]
        x = m.at(p[0], dt)[: rows * C].view(rows, C).float()
        if p[1]:
            add = m.at(p[1], dt)[: period * C].view(period, C).float()
            x = x + add.repeat(rows // period, 1)
        y = F.layer_norm(x, (C,), m.at(p[2], torch.float32)[:C], m.at(p[3], torch.float32)[:C], f[0])
        if i[6] == 1:
            m.at(p[4], dt)[: rows * C].view(rows, C).copy_(split_encode(y))
        else:
            m.at(p[4], dt)[: rows * C].view(rows, C).copy_(y.to(dt))
            if i[6] == 2:
                m.at(p[5], dt)[: rows * C].view(rows, C).copy_(split_encode(y))
    elif k == L.OP_ATTN_ROWS:
        ldq, ldk, ldv, ldo, qoff, koff, voff, ooff, heads, nq, nk, groups, mode, H, W, D = [i[j] for j in range(16)]
        scale = f[0]
        if mode == 0:
            rows = groups * nq
            q = m.at(p[0], dt)[: rows * ldq].view(groups, nq, ldq)[..., qoff:qoff + heads * D].float()
            kx = m.at(p[1], dt)[: groups * nk * ldk].view(groups, nk, ldk)[..., koff:koff + heads * D].float()
            v = m.at(p[2], dt)[: groups * nk * ldv].view(groups, nk, ldv)[..., voff:voff + heads * D].float()
            q = q.view(groups, nq, heads, D).transpose(1, 2); kx = kx.view(groups, nk, heads, D).transpose(1, 2)
            v = v.view(groups, nk, heads, D).transpose(1, 2)
            o = torch.softmax(q @ kx.transpose(2, 3) * scale, -1) @ v
            out = m.at(p[4], dt)[: rows * ldo].view(groups, nq, ldo)
            oo = o.transpose(1, 2).reshape(groups, nq, heads * D)
            out[..., ooff:ooff + heads * D] = split_encode(oo.reshape(rows, heads * D)).view(groups, nq, heads * D) if i[16] else oo.to(dt)
        else:
            wy, wx = (H + 11) // 12, (W + 11) // 12
            B = groups // (wy * wx)
            C = heads * D
            def win(ptr, ld, off, bias):
                t = m.at(ptr, dt)[: B * H * W * ld].view(B, H, W, ld)[..., off:off + C].float()
                pad = torch.zeros(B, wy * 12, wx * 12, C)
                if bias is not None:
                    pad[:] = bias
                pad[:, :H, :W] = t
                pad = pad.view(B, wy, 12, wx, 12, C).permute(0, 1, 3, 2, 4, 5).reshape(B * wy * wx, 144, heads, D)
                return pad.transpose(1, 2)
            kb = m.at(p[5], torch.float32)[:C] if p[5] else None
            vb = m.at(p[6], torch.float32)[:C] if p[6] else None
            q, kx, v = win(p[0], ldq, qoff, None), win(p[1], ldk, koff, kb), win(p[2], ldv, voff, vb)
            o = torch.softmax(q @ kx.transpose(2, 3) * scale, -1) @ v
            o = o.transpose(1, 2).reshape(B, wy, wx, 12, 12, C).permute(0, 1, 3, 2, 4, 5).reshape(B, wy * 12, wx * 12, C)
            out = m.at(p[4], dt)[: B * H * W * ldo].view(B, H, W, ldo)
            oo = o[:, :H, :W]
            out[..., ooff:ooff + C] = split_encode(oo.reshape(B * H * W, C)).view(B, H, W, C) if i[16] else oo.to(dt)
    elif k == L.OP_CHAN_ATTN:
        B, N, C, G = i[0], i[1], i[3], i[4]
        qkv = m.at(p[0], dt)[: B * N * 3 * C].view(B, N, 3, G, C // G).float().permute(2, 0, 3, 4, 1)
        q, kx, v = qkv.unbind(0)
        scale = f[0] if f[0] != 0.0 else N ** -0.5
        o = torch.softmax(q @ kx.transpose(2, 3) * scale, -1) @ v          # [B,G,32,N]
        oo = o.permute(0, 3, 1, 2).reshape(B, N, C)
        m.at(p[4], dt)[: B * N * C].view(B, N, C).copy_(split_encode(oo.reshape(B * N, C)).view(B, N, C) if i[6] else oo.to(dt))
    elif k == L.OP_PROJ_PREP:
        B, N, C = i[0], i[1], i[3]
        x = m.at(p[0], dt)[: B * N * C].view(B, N, C).float()
        v = (x + m.at(p[1], torch.float32)[: N * C].view(N, C)) + m.at(p[2], torch.float32)[:C]
        out = m.at(p[4], dt)[: B * (N + 1) * C].view(B, N + 1, C)
        out[:, 0] = v.mean(1).to(dt)
        out[:, 1:] = v.to(dt)
    elif k == L.OP_ASSEMBLE:
        B, n_img, n_txt, C = i[0], i[1], i[2], i[3]
        out = m.at(p[4], dt)[: B * (n_img + n_txt) * C].view(B, n_img + n_txt, C)
        out[:, :n_img] = m.at(p[0], dt)[: B * n_img * C].view(B, n_img, C)
        out[:, n_img:] = m.at(p[1], dt)[: n_txt * C].view(n_txt, C)
    elif k == L.OP_EMBED_STEP:
        B, C, T, off = i[0], i[3], i[4], i[5]
        st = int(m.at(p[6], torch.int32)[0])
        ids = m.at(p[2], torch.int32)[: B * T].view(B, T)[:, st].long()
        scale = f[0] if f[0] != 0.0 else 1.0
        table = m.at(p[0], dt)
        rows = torch.stack([table[t * C:(t + 1) * C] for t in ids.tolist()]).float() * scale
        pos = m.at(p[1], dt)[(st + off) * C:(st + off + 1) * C].float()
        m.at(p[4], dt)[: B * C].view(B, C).copy_((rows + pos).to(dt))
    elif k == L.OP_ATTN_DECODE:
        ldq, qoff, ldn, koff, voff, ldo, heads, nk_fixed, cap, C, B, ldc = [i[j] for j in range(12)]
        ldc = ldc or C
        q = m.at(p[0], dt)[: B * ldq].view(B, ldq)[:, qoff:qoff + C].float().view(B, heads, 1, 64)
        kc = m.at(p[3], dt)[: B * cap * ldc - 0].view(-1)
        vc = m.at(p[5], dt).view(-1)
        # cache row r of batch b starts at (b*cap + r)*ldc
        def rows(buf, nk):
            n_el = (B * cap - 1) * ldc + C
            t = buf[:n_el]
            idx = (torch.arange(B).view(B, 1) * cap + torch.arange(nk).view(1, nk)) * ldc
            return torch.stack([torch.stack([t[o:o + C] for o in row.tolist()]) for row in idx])   # [B,nk,C]
        if nk_fixed > 0:
            nk = nk_fixed
        else:
            st = int(m.at(p[6], torch.int32)[0]); nk = st + 1
            kn = m.at(p[1], dt)[: B * ldn].view(B, ldn)[:, koff:koff + C]
            vn = m.at(p[2], dt)[: B * ldn].view(B, ldn)[:, voff:voff + C]
            for b in range(B):
                o = (b * cap + st) * ldc
                kc[o:o + C] = kn[b]; vc[o:o + C] = vn[b]
        K = rows(kc, nk).float().view(B, nk, heads, 64).transpose(1, 2)
        Vv = rows(vc, nk).float().view(B, nk, heads, 64).transpose(1, 2)
        o = torch.softmax(q @ K.transpose(2, 3) * f[0], -1) @ Vv
        m.at(p[4], dt)[: B * ldo].view(B, ldo)[:, :C] = o.transpose(1, 2).reshape(B, C).to(dt)
    elif k == L.OP_GREEDY_STEP:
        B, V, ldl, T, max_new, ngram, bos, eos, pad, fbos, feos, inc = [i[j] for j in range(12)]
        step = m.at(p[6], torch.int32)
        st = int(step[0]); cur_len = st + 1
        logits = m.at(p[0], dt)[: B * ldl].view(B, ldl)[:, :V].float().clone()
        if p[1]:
            logits += m.at(p[1], torch.float32)[:V]
        ids = m.at(p[2], torch.int32)[: B * T].view(B, T)
        fin = m.at(p[3], torch.int32)[:B]
        for b in range(B):
            seq = ids[b, :cur_len].tolist()
            if ngram > 0 and cur_len + 1 >= ngram:
                prefix = seq[cur_len - (ngram - 1):]
                for s0 in range(cur_len - ngram + 1):
                    if seq[s0:s0 + ngram - 1] == prefix:
                        logits[b, seq[s0 + ngram - 1]] = float("-inf")
            if fbos >= 0 and cur_len == 1:
                tok = fbos
            elif feos >= 0 and cur_len == max_new:
                tok = feos
            else:
                tok = int(torch.argmax(logits[b]))
            if fin[b]:
                tok = pad
            ids[b, st + 1] = tok
            if not fin[b] and tok == eos:
                fin[b] = 1
        if inc:
            step[0] = st + 1
    elif k == L.OP_CROP_RESIZE:
        from oracle import preprocess_ref as PR
        n, H, W, R, ks, ldo = i[0], i[1], i[2], i[3], i[4], i[13]
        img = m.at(p[0], torch.uint8)[: H * W * 3].view(H, W, 3).numpy()
        boxes = m.at(p[1], torch.int32)[: n * 4].view(n, 4).tolist()
        out = m.at(p[4], dt)[: n * R * R * ldo].view(n, R, R, ldo)
        mean = np.array([f[0], f[1], f[2]], dtype=np.float32); std = np.array([f[3], f[4], f[5]], dtype=np.float32)
        for j, (x0, y0, x1, y1) in enumerate(boxes):
            out[j] = 0
            out[j, :, :, :3] = torch.from_numpy(PR.caption_pixel_values(img, (x0, y0, x1, y1), R, mean, std)).to(dt)
    else:
        raise NotImplementedError(f"interp: op kind {k}")
