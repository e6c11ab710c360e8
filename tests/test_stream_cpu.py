"""BASELINE configs[3] host logic: mixed-resolution stream planning, per-chunk asynchronous gather of packed
element records, prefetching loader — single process and world_size-2 gloo, with a stub parser."""
import os
import socket
import threading

import numpy as np
import torch
import torch.multiprocessing as mp

from omniparser_amd import dist as D
from omniparser_amd import stream as ST


def _elems_for(i, w, h):
    """Deterministic parsed elements of item i (what the stub parser 'finds')."""
    n_text, n_icon = i % 4, 1 + (i * 7) % 9
    el = [{"type": "text", "bbox": [0.01 * j, 0.02, 0.01 * j + 0.05, 0.04], "interactivity": False, "content": f"t{i}-{j}",
           "source": "box_ocr_content_ocr"} for j in range(n_text)]
    el += [{"type": "icon", "bbox": [(j + 1) / 16, (i % 10) / 16, (j + 2) / 16, (i % 10 + 1) / 16], "interactivity": True,
            "content": f"c{j}", "source": "box_yolo_content_yolo"} for j in range(n_icon)]
    rows = [torch.tensor([2, 0, 1000 + i, 2000 + j, w % 1000, h % 1000, 2]) for j in range(n_icon)]
    return el, rows


def _make_parse(sizes, log):
    def parse(frames, ocr):
        shapes = {tuple(f.shape) for f in frames}
        assert len(shapes) == 1 and len(ocr) == len(frames)               # one resolution per device batch
        ids = [int(f[0, 0, 0]) * 256 + int(f[0, 0, 1]) for f in frames]    # id smuggled in the first pixel
        log.append(ids)
        out = [_elems_for(i, *sizes[i]) for i in ids]
        return [o[0] for o in out], [o[1] for o in out]
    return parse


def _make_load(sizes, scale=16):
    def load(i):
        w, h = sizes[i]
        f = torch.zeros(h // scale, w // scale, 3, dtype=torch.uint8)
        f[0, 0, 0], f[0, 0, 1] = i // 256, i % 256
        return f, (None if i % 3 else ([f"o{i}"], [[0, 0, 5, 5]]))
    return load


def _check_records(records, sizes):
    assert records.shape == (len(sizes), D.REC_W)
    for i in range(len(sizes)):
        iid, boxes, conf, cls, cap = D.unpack_record(records[i])
        el, rows = _elems_for(i, *sizes[i])
        assert iid == i and boxes.shape[0] == len(el)
        assert torch.equal(boxes, torch.tensor([e["bbox"] for e in el], dtype=torch.float32))
        assert cls.tolist() == [int(e["type"] == "icon") for e in el] and bool((conf == 1).all())
        icon_rows = [k for k, e in enumerate(el) if e["type"] == "icon"]
        for k, row in zip(icon_rows, rows):
            assert cap[k, : len(row)].tolist() == row.tolist() and int(cap[k, len(row):].sum()) == 0
        for k, e in enumerate(el):
            if e["type"] == "text":
                assert int(cap[k].sum()) == 0


def test_plan_invariants():
    sizes = ST.synthetic_sizes(1581, seed=3)
    assert len({s for s in sizes}) >= 5 and sizes == ST.synthetic_sizes(1581, seed=3)
    for world in (1, 2, 8):
        seen = []
        plans = [ST.plan_stream(sizes, r, world, batch=8, chunk=128) for r in range(world)]
        assert len({len(p) for p in plans}) == 1 and len(plans[0]) == -(-1581 // 128)
        for r, plan in enumerate(plans):
            for ci, ch in enumerate(plan):
                assert (ch.start, ch.n_items) == (plans[0][ci].start, plans[0][ci].n_items)
                firsts = [b[0] for b in ch.batches]
                assert firsts == sorted(firsts)
                for b in ch.batches:
                    assert 1 <= len(b) <= 8 and len({sizes[i] for i in b}) == 1 and b == sorted(b)
                    assert all(i % world == r and ch.start <= i < ch.start + ch.n_items for i in b)
                    seen += b
        assert sorted(seen) == list(range(1581))
    # batching actually happens: far fewer batches than items on one rank
    assert sum(len(ch.batches) for ch in ST.plan_stream(sizes, 0, 1, 8, 128)) < 1581 // 3


def test_single_process_stream_with_prefetch():
    sizes = ST.synthetic_sizes(300, seed=1)
    log, chunks = [], []
    loads, lock = [], threading.Lock()
    base = _make_load(sizes)
    def load(i):
        with lock:
            loads.append(i)
        return base(i)
    out = ST.run_stream(sizes, load, _make_parse(sizes, log), batch=8, chunk=128, prefetch=3,
                        on_chunk=lambda start, rec: chunks.append((start, rec.shape[0])))
    _check_records(out["records"], sizes)
    assert out["items"] == 300 and out["batches"] == len(log) and sorted(loads) == list(range(300))
    assert chunks == [(0, 128), (128, 128), (256, 44)]
    empty = ST.run_stream([], base, _make_parse(sizes, []), batch=8)
    assert empty["records"].shape == (0, D.REC_W) and empty["items"] == 0


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = D.init_from_env("gloo")
    sizes = ST.synthetic_sizes(n, seed=5)
    log = []
    out = ST.run_stream(sizes, _make_load(sizes), _make_parse(sizes, log), rank=r, world=w, batch=4, chunk=32, prefetch=2)
    try:
        _check_records(out["records"], sizes)
        ok = all(i % w == r for b in log for i in b)
    except AssertionError as e:            # report instead of hanging the peer
        ok = repr(e)
    q.put((rank, ok, out["items"]))
    torch.distributed.destroy_process_group()


def test_stream_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n = 75                                  # last chunk is ragged (75 = 2*32 + 11) and odd
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs: p.join(timeout=60)
    assert all(ok is True for _, ok, _ in res), res
    assert sorted(k for _, _, k in res) == [37, 38]
