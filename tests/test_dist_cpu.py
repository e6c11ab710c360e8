"""world_size-2 gloo test of the sharding + packed-record gather (the N>1 path of bench.py)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_items, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from omniparser_amd import dist as D
    r, w, _ = D.init_from_env("gloo")
    mine = D.shard_indices(n_items, r, w)
    recs = []
    for i in mine:
        g = torch.Generator().manual_seed(i)
        k = 1 + (i * 37) % 300
        recs.append(D.pack_record(i, torch.rand(k, 4, generator=g) * 1000, torch.rand(k, generator=g),
                                  torch.randint(0, 5, (k,), generator=g), torch.randint(0, 51289, (k, 21), generator=g)))
    local = torch.stack(recs) if recs else torch.zeros(0, D.REC_W, dtype=torch.int32)
    allr = D.gather_records(local, n_items, r, w)
    ok = True
    for i in range(n_items):
        g = torch.Generator().manual_seed(i)
        k = 1 + (i * 37) % 300
        b, c, cl, cap = (torch.rand(k, 4, generator=g) * 1000, torch.rand(k, generator=g),
                         torch.randint(0, 5, (k,), generator=g), torch.randint(0, 51289, (k, 21), generator=g))
        iid, b2, c2, cl2, cap2 = D.unpack_record(allr[i])
        ok &= iid == i and torch.equal(b, b2) and torch.equal(c, c2) and torch.equal(cl, cl2) and torch.equal(cap, cap2)
    q.put((rank, ok, len(mine)))
    torch.distributed.destroy_process_group()


def test_shard_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_items = 7
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs: p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert sorted(n for _, _, n in res) == [3, 4]


def test_shard_indices_cover_everything():
    from omniparser_amd import dist as D
    for world in (1, 2, 4, 8):
        seen = sorted(i for r in range(world) for i in D.shard_indices(1581, r, world))
        assert seen == list(range(1581))
