"""`-m gpu`, needs >= 2 GPUs (skips cleanly on the 1-GPU box): the N>1 path of bench.py — round-robin shards + ONE all_gather of
packed element records — over RCCL (`backend="nccl"` is RCCL on ROCm) on DEVICE tensors, against the single-rank result.
The CPU twin (gloo) is tests/test_dist_cpu.py.  No scaling curve exists until the driver's SCALE_r*.json stops being skipped."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _record(i, dev):
    from omniparser_amd import dist as D
    g = torch.Generator().manual_seed(i)
    k = 1 + (i * 37) % 300
    return D.pack_record(i, (torch.rand(k, 4, generator=g) * 1000).to(dev), torch.rand(k, generator=g).to(dev),
                         torch.randint(0, 5, (k,), generator=g).to(dev), torch.randint(0, 51289, (k, 21), generator=g).to(dev))


def _worker(rank, world, port, n_items, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    from omniparser_amd import dist as D
    r, w, lr = D.init_from_env("nccl")
    dev = torch.device("cuda", lr)
    mine = D.shard_indices(n_items, r, w)
    local = torch.stack([_record(i, dev) for i in mine]) if mine else torch.zeros(0, D.REC_W, dtype=torch.int32, device=dev)
    allr = D.gather_records(local, n_items, r, w)
    torch.cuda.synchronize(dev)
    single = torch.stack([_record(i, dev) for i in range(n_items)])            # what one rank alone would hold
    q.put((rank, bool(allr.is_cuda and torch.equal(allr, D.gather_records(single, n_items, 0, 1))), len(mine)))
    torch.distributed.destroy_process_group()


def test_shard_and_gather_world2_nccl():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (the driver's multi-GPU node); the gloo twin runs everywhere")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_items = 9
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs: p.join(timeout=120)
    assert all(ok for _, ok, _ in res), res
    assert sorted(n for _, _, n in res) == [4, 5]
