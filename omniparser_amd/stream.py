"""Streamed evaluation over the GPUs of one node (BASELINE configs[3]: an eval set of mixed-resolution
screenshots, sharded over 8 MI355X, parsed elements gathered with RCCL).

The reference evaluates one screenshot at a time in a Python loop on one device
(ref:eval/ss_pro_gpt4o_omniv2.py:37-75 calls `get_som_labeled_img` per sample).  Here the stream is planned up
front from the image sizes alone (headers are cheap to read):

* item i belongs to rank `i % world` — no data-path collective;
* the stream is cut into *chunks* of `chunk` consecutive global ids (default 128).  Every rank walks the same
  chunk sequence, so all ranks issue the same number of collectives: ONE `all_gather` of fixed-width packed
  records per chunk (`dist.REC_W` int32 per screenshot, ≈31 KB), issued asynchronously and completed while the
  next chunk is being parsed — the exchange never stalls the device path;
* inside a rank's part of a chunk, frames of equal size are grouped into device batches of at most `batch`
  frames (one detector plan per (size, batch) — plans are cached LRU, `OMNI_MAX_DETECT_PLANS`), ordered by first
  arrival so the output order is deterministic;
* a host thread pool decodes / uploads batch k+1 while batch k is on the GPU (`prefetch`).

Nothing here touches model code: `parse_fn(frames, ocr) -> (elements per frame, caption-id rows per frame)` is
`ScreenParser.parse_batch(..., return_ids=True)` in production and a stub in the CPU tests.
"""
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import dist as OD

# Synthetic stand-in for the ScreenSpot-Pro resolution mix (dataset absent; SURVEY §8d config 4: "typical
# 2560x1440-3840x2160"): (width, height, share).  Documented as synthetic wherever it is used.
RESOLUTION_MIX = ((2560, 1440, 0.34), (3840, 2160, 0.30), (1920, 1080, 0.12), (2880, 1800, 0.08), (3456, 2234, 0.06),
                  (5120, 2880, 0.05), (2560, 1600, 0.05))


def synthetic_sizes(n: int, seed: int = 0) -> List[Tuple[int, int]]:
    """n (w, h) pairs drawn i.i.d. from RESOLUTION_MIX with a seeded generator."""
    rng = np.random.default_rng(seed)
    p = np.array([m[2] for m in RESOLUTION_MIX])
    pick = rng.choice(len(RESOLUTION_MIX), size=n, p=p / p.sum())
    return [(RESOLUTION_MIX[k][0], RESOLUTION_MIX[k][1]) for k in pick]


@dataclass
class Chunk:
    start: int                       # first global id of the chunk
    n_items: int                     # global ids [start, start + n_items)
    batches: List[List[int]] = field(default_factory=list)     # this rank's device batches (global ids)


def plan_stream(sizes: Sequence[Tuple[int, int]], rank: int, world: int, batch: int = 8, chunk: int = 128) -> List[Chunk]:
    """This rank's schedule.  Invariants (tested): every id of `i % world == rank` appears exactly once; a batch
    holds one resolution and at most `batch` frames; chunk boundaries are identical on every rank."""
    if batch < 1 or chunk < 1:
        raise ValueError("batch and chunk must be >= 1")
    out = []
    for start in range(0, len(sizes), chunk):
        stop = min(start + chunk, len(sizes))
        ch = Chunk(start, stop - start)
        open_by_size: Dict[Tuple[int, int], List[int]] = {}
        for i in range(start, stop):
            if i % world != rank:
                continue
            cur = open_by_size.get(tuple(sizes[i]))
            if cur is None or len(cur) >= batch:
                cur = []
                open_by_size[tuple(sizes[i])] = cur
                ch.batches.append(cur)           # position = first arrival of this group
            cur.append(i)
        out.append(ch)
    return out


def pack_elements(item_id: int, elems: Sequence[dict], caption_rows: Sequence[torch.Tensor]) -> torch.Tensor:
    """Parsed elements of one screenshot -> one `dist.REC_W` record: boxes = element bboxes (ratio xyxy, element
    order), conf = 1, cls = 1 for icon / 0 for text, caption ids on the rows of the captioned (icon, YOLO-sourced)
    elements in caption order."""
    k = min(len(elems), OD.MAX_DET)
    boxes = torch.tensor([e["bbox"] for e in elems[:k]], dtype=torch.float32).reshape(-1, 4)
    cls = torch.tensor([1 if e["type"] == "icon" else 0 for e in elems[:k]], dtype=torch.long)
    cap = torch.zeros(k, OD.CAP_TOK, dtype=torch.long)
    captioned = [i for i, e in enumerate(elems[:k]) if e.get("source") == "box_yolo_content_yolo"]
    for row, i in zip(caption_rows, captioned):
        row = torch.as_tensor(row).view(-1)[: OD.CAP_TOK]
        cap[i, : row.shape[0]] = row
    return OD.pack_record(item_id, boxes, torch.ones(k), cls, cap)


class _PendingGather:
    """One in-flight chunk exchange: padded send buffer, receive buffers and the async work handle."""

    def __init__(self, local: torch.Tensor, ch: Chunk, world: int):
        self.ch, self.world = ch, world
        per = (ch.n_items + world - 1) // world
        self.send = torch.full((per, OD.REC_W), -1, dtype=torch.int32, device=local.device)
        self.send[: local.shape[0]] = local
        if world > 1:
            self.recv = [torch.empty_like(self.send) for _ in range(world)]
            self.work = dist.all_gather(self.recv, self.send, async_op=True)
        else:
            self.recv, self.work = [self.send], None

    def finish(self) -> torch.Tensor:
        if self.work is not None:
            self.work.wait()
        allr = torch.cat(self.recv, 0)
        allr = allr[allr[:, 0] >= 0].cpu()
        out = torch.zeros(self.ch.n_items, OD.REC_W, dtype=torch.int32)
        out[(allr[:, 0] - self.ch.start).long()] = allr
        return out


def run_stream(sizes: Sequence[Tuple[int, int]], load_fn: Callable[[int], Tuple[torch.Tensor, Optional[tuple]]],
               parse_fn: Callable, rank: int = 0, world: int = 1, batch: int = 8, chunk: int = 128, prefetch: int = 2,
               on_chunk: Optional[Callable[[int, torch.Tensor], None]] = None) -> Dict[str, object]:
    """Parse the whole stream.  `load_fn(i) -> (frame uint8 [H,W,3] tensor on the parse device, ocr or None)`.
    Returns {'records': int32 [n, REC_W] ordered by id (all chunks, every rank holds them), 'batches': per-rank
    batch count, 'items': per-rank item count}.  `on_chunk(start, records)` fires as each exchange completes."""
    prefetch = max(int(prefetch), 1)
    plan = plan_stream(sizes, rank, world, batch, chunk)
    todo = [(ci, b) for ci, ch in enumerate(plan) for b in ch.batches]
    done_chunks: List[torch.Tensor] = []
    pending: Optional[_PendingGather] = None
    n_batches = n_items = 0

    def load_batch(ids):
        got = [load_fn(i) for i in ids]
        return [g[0] for g in got], [g[1] if g[1] is not None else ([], []) for g in got]

    def complete(p: _PendingGather):
        rec = p.finish()
        done_chunks.append(rec)
        if on_chunk is not None:
            on_chunk(p.ch.start, rec)

    with ThreadPoolExecutor(max_workers=max(prefetch, 1), thread_name_prefix="omni-load") as pool:
        futs = {k: pool.submit(load_batch, todo[k][1]) for k in range(min(prefetch, len(todo)))}
        k = 0
        for ci, ch in enumerate(plan):
            local = []
            for ids in ch.batches:
                frames, ocr = futs.pop(k).result()
                nxt = k + prefetch
                if nxt < len(todo):
                    futs[nxt] = pool.submit(load_batch, todo[nxt][1])
                elems, rows = parse_fn(frames, ocr)
                for i, el, rw in zip(ids, elems, rows):
                    local.append(pack_elements(i, el, rw))
                n_batches += 1
                n_items += len(ids)
                k += 1
            mine = torch.stack(local) if local else torch.zeros(0, OD.REC_W, dtype=torch.int32)
            if world > 1 and dist.get_backend() == "nccl":      # RCCL moves device buffers
                mine = mine.to(torch.device("cuda", torch.cuda.current_device()), non_blocking=True)
            nxt_pending = _PendingGather(mine, ch, world)      # issue this chunk's exchange ...
            if pending is not None:
                complete(pending)                               # ... and only now wait for the previous one
            pending = nxt_pending
        if pending is not None:
            complete(pending)
    records = torch.cat(done_chunks, 0) if done_chunks else torch.zeros(0, OD.REC_W, dtype=torch.int32)
    return {"records": records, "batches": n_batches, "items": n_items}
