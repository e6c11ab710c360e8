"""Does the STRIDE of the channel-attention operands hold those kernels at 3.5-3.8 TB/s?  OMNI_OP_CHAN_ATTN reads 128-byte pieces (one
32-channel group of q, k or v) out of token rows of 3C floats: 6 KB apart at C = 512.  The same work with C = 32, G = 1 (16 x the
images) reads the same pieces out of 384-byte rows — nearly contiguous streams.  Same kernels, same bytes (16 B per token-channel), only
the stride differs.  `python tools/chan_layout_probe.py` on a GPU box -> one JSON line."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    import torch
    from omniparser_amd import _lib as L
    stream = torch.cuda.Stream()
    out = []
    for (B, N, C) in ((128, 2304, 512), (512, 2304, 128), (2048, 2304, 32), (128, 36864, 128), (512, 36864, 32)):
        G = C // 32
        chunks = (N + 1023) // 1024
        qkv = torch.randn(B * N, 3 * C, device="cuda")
        o = torch.zeros(B * N, C, device="cuda")
        ws = torch.zeros(B * G * chunks * 1024, device="cuda")
        op = L.make_op(L.OP_CHAN_ATTN, L.F32, p=[qkv.data_ptr(), None, None, None, o.data_ptr(), ws.data_ptr()],
                       i={0: B, 1: N, 3: C, 4: G, 5: 1024, 6: 1})
        plan = L.Plan([op])
        plan.run(stream); stream.synchronize()
        ms = min(plan.time(5, stream) for _ in range(3))
        out.append({"B": B, "N": N, "C": C, "row_stride_bytes": 12 * C, "ms": round(ms, 4), "TB_per_s": round(16.0 * B * N * C / ms / 1e9, 3)})
        del qkv, o, ws
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
