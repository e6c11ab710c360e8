"""TEST INFRASTRUCTURE: bench.py's main() on the host emulation of the device library (tests/emu), one process per rank — what
tests/test_bench_world2_cpu.py launches twice with RANK / WORLD_SIZE / MASTER_* set and OMNI_DIST_BACKEND=gloo, so that the N > 1
branches of bench.py (rendezvous, barrier, round-robin shards, gather_records, all_reduce(MAX) of the elapsed time, rank-0-only
JSON line) execute before the driver's multi-GPU run does.  usage: python tests/emu/bench_emulated.py <bench.py arguments>"""
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
for p in (str(ROOT), str(ROOT / "tests"), str(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    import torch
    import gpu_checks  # noqa: F401  (emulated_device redirects its DEV / _sync when present)
    from emu_runtime import emulated_device
    saved = {k: getattr(torch.cuda, k) for k in ("set_device", "max_memory_allocated", "empty_cache")}
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.max_memory_allocated = lambda *a, **k: 0
    torch.cuda.empty_cache = lambda: None
    try:
        with emulated_device():
            import bench
            bench.main()
    finally:
        for k, v in saved.items():
            setattr(torch.cuda, k, v)


if __name__ == "__main__":
    main()
