"""One process per GPU; screenshots are independent units, sharded round-robin.

The only exchange step of the path is the batched-eval gather: every rank packs its parsed elements
into fixed-width records and ONE all_gather per chunk moves them (RCCL over xGMI on GPUs — backend
"nccl" is RCCL on ROCm; "gloo" in CPU tests).  ~31 KB per screenshot: latency-bound, so it is never
issued per screenshot.
"""
import os
from typing import List

import torch
import torch.distributed as dist

MAX_DET = 300
CAP_TOK = 21          # decoder_start + 20 generated ids (ref:util/utils.py:125 max_new_tokens=20)
REC_W = 2 + MAX_DET * (4 + 1 + 1 + CAP_TOK)   # id, K, boxes, conf, cls, caption ids (as f32-bitcast i32)


def init_from_env(backend: str = None):
    """Returns (rank, world, local_rank).  No-op for single-process runs."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        # OMNI_DIST_BACKEND: the CPU tests run the N > 1 path of bench.py on gloo (tests/test_bench_world2_cpu.py)
        backend = backend or os.environ.get("OMNI_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def self_launch(n: int, argv=None, env_extra=None) -> int:
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): the script owns its fan-out.  The reference has no launcher to
    match (ref:util/omniparser.py:16-32 is one process on one device), so a plain invocation must work the way
    `python -m torch.distributed.run --nproc-per-node N` does: N children of this very command line, one per GPU, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (127.0.0.1, a free port); stdout and stderr are inherited, so rank 0's
    one JSON line is this process's stdout.  The parent touches no GPU.  Returns the largest child exit code; if a rank dies, the
    others (exact PIDs, nothing by pattern) are terminated so a failed rendezvous cannot hang the job."""
    import socket
    import subprocess
    import sys
    import time
    argv = list(sys.argv if argv is None else argv)
    # The rendezvous port: found by binding port 0.  Rounds 5 closed that socket before the children started, which left a window in
    # which another process could take the port (advisor, round 5).  The socket now stays BOUND (never listening, SO_REUSEADDR) until
    # the ranks have exited: Linux lets rank 0's store — which sets SO_REUSEADDR as well — bind and listen on the same address while no
    # other LISTENING socket holds it, and routes connections to the listener only; nobody else can be handed the port by bind(0).
    holder = socket.socket()
    holder.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    holder.bind(("127.0.0.1", 0))
    port = holder.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMNI_SELF_LAUNCHED="1", **(env_extra or {}))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this driver (RCCL across processes)
        procs.append(subprocess.Popen([sys.executable, *argv], env=env))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0 and rc == 0:
                rc = r if r > 0 else 128 - r
                deadline = time.time() + 10.0
                for q in live:                                      # one rank failed: the others would wait in a collective forever
                    q.terminate()
                for q in live:
                    try:
                        q.wait(max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        q.kill()
        time.sleep(0.05)
    holder.close()
    return rc


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """round-robin: item i belongs to rank i % world."""
    return list(range(rank, n_items, world))


def pack_record(item_id: int, boxes: torch.Tensor, conf: torch.Tensor, cls: torch.Tensor,
                caption_ids: torch.Tensor = None) -> torch.Tensor:
    """-> int32[REC_W] on the tensors' device (floats bit-cast, no precision loss)."""
    dev = boxes.device
    k = min(int(boxes.shape[0]), MAX_DET)
    rec = torch.zeros(REC_W, dtype=torch.int32, device=dev)
    rec[0] = item_id
    rec[1] = k
    o = 2
    rec[o:o + 4 * k] = boxes[:k].contiguous().float().view(torch.int32).flatten()
    o += 4 * MAX_DET
    rec[o:o + k] = conf[:k].contiguous().float().view(torch.int32)
    o += MAX_DET
    rec[o:o + k] = cls[:k].to(torch.int32)
    o += MAX_DET
    if caption_ids is not None and caption_ids.numel():
        kk = min(k, caption_ids.shape[0])
        t = min(CAP_TOK, caption_ids.shape[1])
        blk = torch.zeros(MAX_DET, CAP_TOK, dtype=torch.int32, device=dev)
        blk[:kk, :t] = caption_ids[:kk, :t].to(torch.int32)
        rec[o:] = blk.flatten()
    return rec


def unpack_record(rec: torch.Tensor):
    rec = rec.cpu()
    item_id, k = int(rec[0]), int(rec[1])
    o = 2
    boxes = rec[o:o + 4 * k].view(torch.float32).view(k, 4)
    o += 4 * MAX_DET
    conf = rec[o:o + k].view(torch.float32)
    o += MAX_DET
    cls = rec[o:o + k].long()
    o += MAX_DET
    cap = rec[o:].view(MAX_DET, CAP_TOK)[:k].long()
    return item_id, boxes, conf, cls, cap


def gather_records(local: torch.Tensor, n_items: int, rank: int, world: int) -> torch.Tensor:
    """local: int32[n_local, REC_W] (this rank's shard, any order).  Returns int32[n_items, REC_W]
    ordered by item id on every rank.  One all_gather; shards are padded to equal length."""
    if world == 1:
        out = torch.zeros(n_items, REC_W, dtype=torch.int32, device=local.device)
        out[local[:, 0].long()] = local
        return out
    per = (n_items + world - 1) // world
    pad = torch.full((per, REC_W), -1, dtype=torch.int32, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    allr = torch.cat(bufs, 0)
    allr = allr[allr[:, 0] >= 0]
    out = torch.zeros(n_items, REC_W, dtype=torch.int32, device=local.device)
    out[allr[:, 0].long()] = allr
    return out
