"""Algorithmic work of one plan op (`omni_op_t`): FLOPs, bytes, a shape key and the kernel family that executes it.

"Algorithmic" = every operand read once and every result written once (weights included) — the traffic a perfectly cached execution
of the op list needs; FLOPs are 2 x MAC for GEMMs / attention.  `bench.py::roofline` divides these by the HIP-event time of the
launches (`omni_plan_profile`) for its `achieved` figure and its `per_kernel` list; `tools/plan_table.py` and
`tools/caption_profile.py` print tables from the same functions, so the bench line and the committed tables cannot disagree.
Slot meanings: include/omni_amd.h."""

NAMES = {1: "conv_igemm/conv_split", 2: "avgpool2", 3: "maxpool", 4: "resize_nearest", 5: "letterbox", 6: "detect_decode", 7: "nms",
         8: "dwconv3", 9: "layernorm", 10: "attn_rows (window / MHA)", 11: "chan_attn", 12: "proj_prep", 13: "assemble", 14: "embed_step",
         15: "attn_decode", 16: "greedy_step", 17: "crop_resize", 18: "dwconv3_ln", 19: "split_convert", 20: "hand_off", 21: "overlay",
         22: "png_pack", 23: "png_deflate", 24: "mlp_fused"}

MFMA_KINDS = (1, 24, 10)          # priced against the dense f16 MFMA peak; everything else against HBM


def op_work(op, esz=4):
    """(flops, algorithmic bytes) of one op."""
    i = op.i
    k = op.kind
    if k == 1:
        M, N, K = i[0] * i[10] * i[11], i[12], i[6] * i[7] * i[3]
        if i[25]:                                   # row-patch mode: algorithmic K = k x k taps x 4 stored channels (the op's 32 k holds zero weights)
            K = i[6] * i[6] * i[4]
            return 2 * M * N * K, esz * (i[0] * i[1] * i[2] * i[4] + N * K + M * N)
        return 2 * M * N * K, esz * (i[0] * i[1] * i[2] * i[3] + N * K + M * N * (2 if op.p[3] else 1))
    if k == 24:                                     # fc1 + GELU + fc2 + residual: h in, residual in, y out, both weight matrices
        rows, C, hid = i[0] * max(i[1], 1), i[3], i[12]
        return 4 * rows * C * hid, esz * (3 * rows * C + 2 * C * hid)
    if k in (2, 3, 4):
        return 0, esz * (i[0] * i[1] * i[2] * i[3] + i[0] * max(i[10], 1) * max(i[11], 1) * i[3])
    if k in (8, 18):
        n = i[0] * i[1] * i[2] * i[3]
        return 18 * n, esz * n * (2 if k == 8 else 3)
    if k == 9:
        n = i[0] * max(i[1], 1) * i[3]
        return 8 * n, esz * 2 * n
    if k == 10:
        heads, nq, nk, groups, D = i[8], i[9], i[10], i[11], i[15]
        return 4 * groups * heads * nq * nk * D, esz * groups * (nq + 2 * nk + nq) * heads * D
    if k == 11:
        B, N, C = i[0], i[1], i[3]
        return 4 * B * N * C * 32, esz * B * N * C * 4
    if k == 15:
        B, heads, nk = i[10], i[6], (i[7] if i[7] > 0 else i[8])
        return 4 * B * heads * nk * 64, esz * B * nk * i[9] * 2
    if k == 16:
        return 0, esz * i[0] * i[1]
    return 0, 0


def op_shape(op):
    """(rows, channels out, K or a variant tag): the key launches are grouped by."""
    i, k = op.i, op.kind
    if k in (1, 19):
        return (i[0] * max(i[10], 1) * max(i[11], 1) if k == 1 else i[0] * max(i[1], 1), i[12] if k == 1 else i[3], i[6] * i[7] * i[3] if k == 1 else 0)
    if k == 24:
        return (i[0] * max(i[1], 1), i[3], i[12])
    if k in (8, 18):
        return (i[0] * i[1] * i[2], i[3], 0)
    if k == 9:
        return (i[0] * max(i[1], 1), i[3], i[6])
    if k == 10:
        return (i[11] * i[9], i[8] * i[15], i[12])
    if k == 11:
        return (i[0] * i[1], i[3], 0)
    if k == 15:
        return (i[10], i[9], i[7] if i[7] > 0 else i[8])
    return (i[0], i[3], 0)


def op_kernel(op):
    """the kernel family a launch of this op runs (kind 1 is served by three kernels, chosen by the planner: slot i20)."""
    if op.kind == 1:
        return "gemm_dma" if op.i[20] == 2 else ("conv_split" if op.i[20] else "conv_igemm")
    return NAMES.get(op.kind, str(op.kind))
