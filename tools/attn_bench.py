"""Time the window-attention op at the DaViT stage shapes (HIP events)."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    import torch
    from omniparser_amd import _lib as L
    stream = torch.cuda.Stream()
    B = int(os.environ.get("BATCH", "128"))
    for (H, heads, name) in ((192, 4, "stage0"), (96, 8, "stage1"), (48, 16, "stage2"), (24, 32, "stage3")):
        C = heads * 32
        qkv = torch.randn(B * H * H, 3 * C, device="cuda")
        bias = torch.randn(3 * C, device="cuda")
        o = torch.zeros(B * H * H, C, device="cuda")
        nw = ((H + 11) // 12) ** 2
        op = L.make_op(L.OP_ATTN_ROWS, L.F32, p=[qkv.data_ptr(), qkv.data_ptr(), qkv.data_ptr(), None, o.data_ptr(),
                                                 bias.data_ptr() + 4 * C, bias.data_ptr() + 8 * C],
                       i={0: 3 * C, 1: 3 * C, 2: 3 * C, 3: C, 4: 0, 5: C, 6: 2 * C, 7: 0, 8: heads, 9: 144, 10: 144, 11: B * nw,
                          12: 1, 13: H, 14: H, 15: 32}, f={0: 32 ** -0.5})
        plan = L.Plan([op])
        plan.run(stream); stream.synchronize()
        ms = plan.time(3, stream)
        flops = 4.0 * B * nw * heads * 144 * 144 * 32
        print(f"{name}: H={H} heads={heads}  {ms:8.3f} ms   {flops/ms/1e9:7.2f} TF/s (QK^T+PV)")
        del qkv, o


if __name__ == "__main__":
    main()
