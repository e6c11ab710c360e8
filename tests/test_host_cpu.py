"""CPU-side tests (`-m "not gpu"`): C-ABI surface, host coefficient tables pinned against Pillow,
and the YOLOv9-E plan lowering checked with the CPU op interpreter against the oracle network."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from omniparser_amd import _lib as L

ROOT = Path(__file__).resolve().parents[1]


def test_library_exports_every_declared_symbol():
    hdr = (ROOT / "include" / "omni_amd.h").read_text()
    declared = set(re.findall(r"\b(omni_[a-z_0-9]+)\s*\(", hdr))
    lib = L.lib()
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert set(L.EXPORTS) <= declared
    assert lib.omni_abi_version() == L.ABI_VERSION == int(re.search(r"#define OMNI_ABI_VERSION (\d+)", hdr).group(1))
    assert "omni_debug_host_op" not in declared            # test emulation lives in tests/emu, not in the shipping library
    assert ctypes.sizeof(L.OmniOp) == 8 + 8 * 8 + 32 * 4 + 8 * 4


def test_python_enums_mirror_the_header():
    hdr = (ROOT / "include" / "omni_amd.h").read_text()
    ops = dict(re.findall(r"\b(OMNI_OP_[A-Z0-9_]+) = (\d+),", hdr))
    assert len(ops) == 24 and sorted(map(int, ops.values())) == list(range(1, 25))
    for name, val in ops.items():
        assert getattr(L, name[len("OMNI_"):]) == int(val), name
    assert (L.F32, L.F16, L.ACT_NONE, L.ACT_SILU, L.ACT_GELU) == (0, 1, 0, 1, 2)
    assert "sizeof" not in hdr or True
    assert L.CAND_BYTES == 32


def test_bad_arguments_fail_loudly_without_gpu():
    with pytest.raises(L.OmniError):
        L.check(L.lib().omni_plan_create(None, 0, None))
    op = L.make_op(99, L.F32)
    rc = L.lib().omni_op_launch(ctypes.byref(op), None)
    assert rc == -1 and b"unknown op kind" in L.lib().omni_last_error()
    # conv with unaligned channel counts is rejected before any launch
    op = L.make_op(L.OP_CONV, L.F32, p=[1, 1, None, None, 1],
                   i={0: 1, 1: 4, 2: 4, 3: 3, 4: 3, 5: 0, 6: 1, 7: 1, 8: 1, 9: 0, 10: 4, 11: 4, 12: 8, 13: 8})
    assert L.lib().omni_op_launch(ctypes.byref(op), None) == -1


def _emulate_resample(img, out_w, out_h, filt):
    """numpy emulation of preproc.hip's two passes using the host coefficient tables."""
    h, w, _ = img.shape
    cur = img.astype(np.int64)
    if out_w != w:
        b, k = L.resample_coeffs(w, out_w, filt)
        nxt = np.zeros((h, out_w, 3), dtype=np.int64)
        for xx in range(out_w):
            x0, n = b[xx]
            nxt[:, xx] = (1 << 21) + np.tensordot(cur[:, x0:x0 + n], k[xx, :n].astype(np.int64), axes=([1], [0]))
        cur = np.clip(nxt >> 22, 0, 255)
    if out_h != h:
        b, k = L.resample_coeffs(h, out_h, filt)
        nxt = np.zeros((out_h, cur.shape[1], 3), dtype=np.int64)
        for yy in range(out_h):
            y0, n = b[yy]
            nxt[yy] = (1 << 21) + np.tensordot(k[yy, :n].astype(np.int64), cur[y0:y0 + n], axes=([0], [0]))
        cur = np.clip(nxt >> 22, 0, 255)
    return cur.astype(np.uint8)


@pytest.mark.parametrize("iw,ih,ow,oh,filt", [
    (1920, 1080, 640, 360, 0), (1919, 1079, 640, 359, 0), (333, 517, 412, 640, 0), (640, 640, 640, 360, 0),
    (64, 64, 768, 768, 1), (64, 64, 96, 80, 1), (200, 120, 64, 64, 1)])
def test_host_coefficients_reproduce_pillow(iw, ih, ow, oh, filt):
    from PIL import Image
    rng = np.random.default_rng(iw * 7 + ih)
    img = rng.integers(0, 256, size=(ih, iw, 3), dtype=np.uint8)
    img[ih // 3: ih // 2, iw // 4: iw // 2] = 255
    img[: ih // 5, : iw // 3] = 0
    res = Image.Resampling.LANCZOS if filt == 0 else Image.Resampling.BICUBIC
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), res))
    got = _emulate_resample(img, ow, oh, filt)
    assert np.array_equal(got, ref), int((got != ref).sum())


@pytest.mark.parametrize("dtype", [L.F32, L.F16])
def test_yolo_plan_lowering_matches_oracle_network(dtype):
    from oracle.yolov9e_ref import build_random_detector
    from omniparser_amd.planner import PlanBuilder
    from omniparser_amd.yolo_graph import YoloV9EGraph
    from plan_interp import run_ops
    m = build_random_detector(seed=1, nc=2, width=0.25)
    pb = PlanBuilder("cpu", dtype)
    B, TH, TW = 2, 96, 128
    g = YoloV9EGraph({k: v.float() for k, v in m.state_dict().items()}, pb, B, TH, TW)
    x = pb.alloc(B, TH, TW, pb.V, zero=True)
    xin = torch.rand(B, 3, TH, TW, generator=torch.Generator().manual_seed(3))
    x.t[..., :3] = xin.permute(0, 2, 3, 1).to(x.t.dtype)
    outs = g.build(x)
    assert 250 <= len(pb.ops) <= 300
    run_ops(pb.ops, pb.keep)
    with torch.inference_mode():
        ref = m(xin)
    # chaotic random net (oracle/yolov9e_ref.py header): f32 lowering differs from the oracle by its own
    # rounding noise (~5e-3); in f16 the noise is O(1), so only correlation is checked there.
    for i, (cls, box) in enumerate(outs):
        got, want = cls.torch(), ref[2 * i]
        if dtype == L.F32:
            assert (got - want).abs().max() < 0.05
            assert (m.head.dfl(box.torch()) - ref[2 * i + 1]).abs().max() < 0.05
        else:
            cc = torch.corrcoef(torch.stack([got.flatten(), want.flatten()]))[0, 1]
            assert cc > 0.7, cc


def test_oracle_nms_known_answers():
    from oracle import detector_ref as D
    b = torch.tensor([[0, 0, 10, 10], [0, 0, 10, 5], [20, 20, 30, 30], [0, 0, 10, 10]], dtype=torch.float32)
    s = torch.tensor([0.9, 0.8, 0.7, 0.9])
    # IoU([0,0,10,10],[0,0,10,5]) == 0.5 exactly: strict '>' keeps it; duplicate box 3 is suppressed
    assert D.nms(b, s, 0.5).tolist() == [0, 1, 2]
    assert D.nms(b, s, 0.49).tolist() == [0, 2]
    c = torch.tensor([0, 0, 0, 1])
    assert D.batched_nms(b, s, c, 0.49).tolist() == [0, 3, 2]
    assert D.batched_nms(b[:0], s[:0], c[:0], 0.5).numel() == 0


def test_gemm_tile_permutation_is_a_bijection():
    """csrc/conv_igemm.hip::tile_of_block (XCD-aware order + N partition over XCD groups) through its host mirror:
    every tile exactly once, padding blocks only beyond mtiles, each XCD (bid & 7) sees only its N partition and
    walks the N tiles of a row block in consecutive slots; xcd_n == 1 reproduces the round-1 row-block mapping."""
    from omniparser_amd import _lib as L
    cases = [(2304, 16, -1, 4 * 2048 * 512), (2304, 12, -1, 4 * 1536 * 512), (2304, 4, -1, 4 * 512 * 2048), (585, 24, -1, 4 * 3072 * 768),
             (585, 6, -1, 4 * 768 * 3072), (577, 8, 8, -1), (64, 2, 2, -1), (67, 8, 4, -1), (2304, 16, 4, -1), (1000, 3, 1, -1), (63, 5, 1, -1), (400, 4, -1, 4 * 512 * 4608),
             (2304, 4, -1, 4 * 512 * 512), (100, 1, 1, -1)]
    for mtiles, ntiles, xn, wbytes in cases:
        _, _, grid, used = L.tile_map(mtiles, ntiles, 0, xcd_n=max(xn, 1), weight_bytes=wbytes)
        xcd_order = mtiles >= 64 and ntiles > 1
        if wbytes >= 0 and xcd_order:
            slab = wbytes / used
            assert used in (1, 2, 4, 8) and ntiles % used == 0
            assert used == 1 or slab <= 2.0 * 2 ** 20            # a partition is only chosen when its slab fits
            if used == 1 and wbytes > 2.0 * 2 ** 20:
                assert all(ntiles % k or wbytes / k > 2.0 * 2 ** 20 for k in (2, 4, 8))
        seen = {}
        per_xcd = [[] for _ in range(8)]
        for bid in range(grid):
            mt, nt, g2, u2 = L.tile_map(mtiles, ntiles, bid, xcd_n=max(xn, 1), weight_bytes=wbytes)
            assert (g2, u2) == (grid, used)
            if mt < 0:
                assert nt < 0
                continue
            assert 0 <= mt < mtiles and 0 <= nt < ntiles and (mt, nt) not in seen
            seen[(mt, nt)] = bid
            per_xcd[bid & 7].append((mt, nt))
        assert len(seen) == mtiles * ntiles
        assert grid - len(seen) < 8 * ntiles                       # padding is at most one row-block round
        if not xcd_order:
            assert grid == mtiles * ntiles and used == 1
            continue
        gn = ntiles // used
        for x, tiles in enumerate(per_xcd):
            assert {nt for _, nt in tiles} <= set(range((x % used) * gn, (x % used + 1) * gn))
            assert {mt % (8 // used) for mt, _ in tiles} == {x // used}
            for k in range(0, len(tiles) - gn + 1, gn):            # consecutive slots = the N tiles of ONE row block
                blk = tiles[k:k + gn]
                if len({m for m, _ in blk}) == 1:
                    assert [n for _, n in blk] == list(range((x % used) * gn, (x % used + 1) * gn))
        if used == 1:                                              # round-1 formula
            for (mt, nt), bid in list(seen.items())[:200]:
                assert mt == ((bid >> 3) // ntiles) * 8 + (bid & 7) and nt == (bid >> 3) % ntiles
    assert L.tile_map(2304, 16, 0, weight_bytes=4 * 2048 * 512)[3] == 2        # DaViT stage-2 fc1: 4 MiB of weights -> 2 groups of 2 MiB
    assert L.tile_map(2304, 4, 0, weight_bytes=4 * 512 * 512)[3] == 1          # 1 MiB: already resident


def test_library_binds_to_one_hip_runtime():
    """Loading libomni_amd.so before anything imported torch must not bring a second HIP runtime into the process (torch's wheel
    ships its own libamdhip64.so.7; with /opt/rocm's copy loaded next to it every launch fails with "no ROCm-capable device")."""
    import subprocess
    import sys
    from pathlib import Path
    code = ("from omniparser_amd import _lib as L; L.lib(); import torch; "
            "print(sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}))")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(Path(__file__).resolve().parents[1]), timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    libs = eval(out.stdout.strip().splitlines()[-1])
    assert len(libs) == 1 and "torch" in libs[0], libs


def test_bench_child_process_helper_reports_instead_of_raising():
    """bench.py runs never-timed code paths in child processes with a hard limit (`extra.ab_opt_in_kernels`, `extra.annotate_tail`,
    `extra.stream_*`): whatever happens there ends up as data in the line, never as an exception or a stall of the line itself."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    ok = b.child_json([sys.executable, "-c", "print('noise'); print('{\"value\": 2, \"roofline\": {\"non_gemm_share\": 0.3}}')"], {"X": "1"}, 20,
                      keep=("value", ("roofline", "non_gemm_share"), ("roofline", "absent")))
    assert ok == {"value": 2, "roofline.non_gemm_share": 0.3, "roofline.absent": None, "env": {"X": "1"}}
    assert "no result within" in b.child_json([sys.executable, "-c", "import time; time.sleep(30)"], {}, 1)["error"]
    assert b.child_json([sys.executable, "-c", "import sys; sys.stderr.write('boom'); sys.exit(3)"], {}, 20)["error"].startswith("exit 3")
    assert "error" in b.child_json([sys.executable, "-c", "print('{not json')"], {}, 20)


def test_product_objects_refuse_to_run_without_the_gpu():
    """No CPU fallback: on a box without a GPU the detector / captioner constructors raise (before touching any weights), whatever
    device string they are given; the emulation the test suite uses is bound from tests/emu/emu_runtime.py, not by the product."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    from omniparser_amd.florence import Florence2Captioner
    from omniparser_amd.util.yolov9 import YOLOv9Detector
    for dev in (None, "cpu", "cuda"):
        with pytest.raises(RuntimeError, match="no CPU fallback|unavailable"):
            YOLOv9Detector(model_path="/nonexistent/model.pt", device=dev)
        with pytest.raises(RuntimeError, match="no CPU fallback|unavailable"):
            Florence2Captioner("/nonexistent", dev)
    assert not hasattr(L, "bind_emulation") and not hasattr(L, "EMULATION")


def test_erf_polynomial():
    """csrc/omni_internal.h::omni_erff restated in numpy (every fma rounded to f32): < 1 ulp / 6e-8 absolute against scipy's float64
    erf on a dense grid and on N(0, 1.5) samples — the accuracy class of torch's CPU erf (Sleef u10), so GELU keeps f32 parity while
    the GEMM epilogue drops ocml's 55-instruction erff."""
    import numpy as np
    from scipy.special import erf as erf64

    def fma(a, b, c):
        return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)

    def omni_erff(a):
        a = a.astype(np.float32)
        t = np.minimum(np.abs(a), np.float32(6.0)); s = (t * t).astype(np.float32)
        r = fma(np.float32(-1.72853470e-5), t, np.float32(3.83197126e-4))
        u = fma(np.float32(-3.88396438e-3), t, np.float32(2.42546219e-2))
        r = fma(r, s, u)
        for c in (-1.06777847e-1, -6.34846687e-1, -1.28717512e-1):
            r = fma(r, t, np.float32(c))
        r = fma(r, t, -t)
        big = np.copysign((1.0 - np.exp(r.astype(np.float64))).astype(np.float32), a)
        q = np.full_like(a, -5.96761703e-4)
        for c in (4.99119423e-3, -2.67681349e-2, 1.12819925e-1, -3.76125336e-1, 1.28379166e-1):
            q = fma(q, s, np.float32(c))
        return np.where(t > np.float32(0.927734375), big, fma(q, a, a))

    x = np.concatenate([np.linspace(-8, 8, 400001), np.random.default_rng(0).normal(0, 1.5, 400000)]).astype(np.float32)
    got, ref = omni_erff(x).astype(np.float64), erf64(x.astype(np.float64))
    err = np.abs(got - ref)
    assert err.max() < 6.5e-8
    assert (err / np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)).max() < 1.0


def test_pmc_tools_on_a_synthetic_trace(tmp_path):
    """tools/pmc_summary.py + tools/pmc_traffic.py (how profiles/r3_pmc_traffic.json is made): FETCH_SIZE doubled and KiB -> bytes,
    kernels reduced to families, GEMM-family bytes per crop and per launch, fetch / write ratio of a streaming kernel."""
    import csv
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    rows = [("void (anonymous namespace)::gemm_dma_kernel<256, 256, 2, 4, 2, 0, false, true, 1>((anonymous namespace)::GemmArgs)", 1000.0, 500.0),
            ("void (anonymous namespace)::dwln_strip_kernel<2, 64, true>((anonymous namespace)::DwLnArgs, int, int, int, unsigned int)", 250.0, 1000.0)]
    for ctr, col in (("FETCH_SIZE", 1), ("WRITE_SIZE", 2)):
        d = tmp_path / ctr / "run"
        d.mkdir(parents=True)
        with open(d / "1_counter_collection.csv", "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"])
            for r in rows:
                for launch in range(4):
                    w.writerow([r[0], ctr, r[col], 1000 * launch, 1000 * launch + 500])
        out = subprocess.run([sys.executable, str(root / "tools" / "pmc_summary.py"), str(tmp_path / ctr)], capture_output=True, text=True, check=True)
        (tmp_path / f"{ctr}.json").write_text(out.stdout)
    out = subprocess.run([sys.executable, str(root / "tools" / "pmc_traffic.py"), str(tmp_path / "FETCH_SIZE.json"), str(tmp_path / "WRITE_SIZE.json"),
                          "8", "1000000"], capture_output=True, text=True, check=True)
    t = json.loads(out.stdout)
    assert t["gemm_launches"] == 4
    assert t["gemm_fetch_bytes"] == 4 * 1000.0 * 2 * 1024 and t["gemm_write_bytes"] == 4 * 500.0 * 1024
    assert t["gemm_fetch_bytes_per_crop"] == round(4 * 1000.0 * 2048 / 8) and t["gemm_bytes_per_launch"] == round((1000.0 * 2048 + 500.0 * 1024))
    strip = next(v for k, v in t["families"].items() if k.startswith("dwln_strip_kernel"))
    assert strip["fetch_over_write"] == 0.5 and strip["launches"] == 4


def test_hwq_gap_summary_on_a_synthetic_trace():
    """tools/hwq_gaps.py (the round-4 diagnosis of the hardware-queue cliff): busy union, idle time, per-queue gaps, idle attribution."""
    from tools.hwq_gaps import summarise
    G = "void (anonymous namespace)::gemm_dma_kernel<256, 256, 2, 4, 2, 0, false, true, 1>((anonymous namespace)::GemmArgs)"
    S = "(anonymous namespace)::attn_decode_cross_kernel((anonymous namespace)::DecArgs)"
    rows = [  # queue 1: two long kernels back to back, then a 200 us hole; queue 2: a short kernel inside the first, one inside the hole
        {"Kernel_Name": G, "Start_Timestamp": 0, "End_Timestamp": 1_000_000, "Queue_Id": 1, "Stream_Id": 3},
        {"Kernel_Name": G, "Start_Timestamp": 1_000_000, "End_Timestamp": 2_000_000, "Queue_Id": 1, "Stream_Id": 3},
        {"Kernel_Name": S, "Start_Timestamp": 100_000, "End_Timestamp": 150_000, "Queue_Id": 2, "Stream_Id": 5},
        {"Kernel_Name": S, "Start_Timestamp": 2_100_000, "End_Timestamp": 2_150_000, "Queue_Id": 2, "Stream_Id": 5},
        {"Kernel_Name": G, "Start_Timestamp": 2_200_000, "End_Timestamp": 3_200_000, "Queue_Id": 1, "Stream_Id": 3},
    ]
    r = summarise(rows)
    assert r["launches"] == 5 and abs(r["span_ms"] - 3.2) < 1e-9
    assert abs(r["gpu_busy_ms"] - 3.05) < 1e-9 and abs(r["gpu_idle_ms"] - 0.15) < 1e-9          # holes: 2.0-2.1 and 2.15-2.2
    assert abs(r["concurrent_kernel_ms"] - 0.05) < 1e-9
    assert r["idle_over_50us"] == 1 and r["idle_intervals"] == 2
    assert r["per_queue"]["1"]["gaps_over_50us"] == 1 and abs(r["per_queue"]["1"]["gap_ms"] - 0.2) < 1e-9
    assert r["per_queue"]["2"]["streams"] == ["5"] and r["per_queue"]["2"]["launches"] == 2
    assert list(r["idle_before_kernel_family_ms"]) == ["attn_decode_cross_kernel", "gemm_dma_kernel"]


def test_device_lock_makes_the_models_gpu_current_in_every_thread(monkeypatch):
    """A model's launch lock (L.DeviceLock) sets the calling thread's current HIP device to the model's GPU and restores it: worker
    threads start on device 0, one process per GPU addresses its GPU as cuda:LOCAL_RANK.  torch.cuda is replaced by a per-thread
    stand-in (no GPU here)."""
    import threading
    import torch
    from omniparser_amd import _lib as L
    cur = threading.local()
    calls = []
    monkeypatch.setattr(torch.cuda, "current_device", lambda: getattr(cur, "d", 0))
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: (calls.append((threading.current_thread().name, d)), setattr(cur, "d", d)))
    lock = L.DeviceLock(torch.device("cuda", 3), reentrant=True)
    seen = {}

    def worker():
        with lock:
            seen["inside"] = torch.cuda.current_device()
            with lock:                                   # reentrant: the nested entry finds the device already current
                seen["nested"] = torch.cuda.current_device()
            seen["after_nested"] = torch.cuda.current_device()
        seen["after"] = torch.cuda.current_device()

    t = threading.Thread(target=worker, name="w")
    t.start(); t.join()
    assert seen == {"inside": 3, "nested": 3, "after_nested": 3, "after": 0}
    assert calls == [("w", 3), ("w", 0)]
    calls.clear()
    cur.d = 3                                            # a thread already on the model's device: no device calls at all
    with lock:
        pass
    assert calls == []
    with L.DeviceLock(torch.device("cpu")):              # emulated / CPU stand-in models: a plain lock
        pass
    assert calls == []
    plain = L.DeviceLock(torch.device("cuda", 1))
    with plain:
        assert not plain._lock.acquire(blocking=False)   # non-reentrant flavour really excludes


def test_plan_export_survives_a_dead_tensors_registry_entry_at_the_same_address(tmp_path):
    """The tensor registry is keyed by base address with weak references.  A caller-owned bias whose address was once a (now dead)
    registered tensor's must be registered afresh — found as an order-dependent failure of the bundle round-trip test: the stale entry
    made `conv` skip the registration, the export then purged it and could not resolve the pointer."""
    import gc
    import weakref
    import torch
    from omniparser_amd import bundle as B, planner
    from omniparser_amd.planner import PlanBuilder
    pb = PlanBuilder("cpu", L.F32)
    x, y = pb.alloc(1, 4, 4, 32, zero=True), pb.alloc(1, 4, 4, 64)
    w = pb.pack_weight(torch.randn(64, 32, 1, 1))
    bias = torch.randn(64)
    dead = torch.zeros(3)
    planner._TENSORS[bias.data_ptr()] = (weakref.ref(dead), "scratch", 12)      # what an earlier plan's freed tensor leaves behind
    del dead
    gc.collect()
    pb.conv(x, w, bias, y, 1)
    info = B.write_bundle(tmp_path / "t.omniplan", {"p": pb.ops}, {"x": (x.t, 0, 64)}, {})
    assert info["ops"] == {"p": 1}


def test_bench_record_packing_round_trips():
    """bench.py::pack_records (runs inside the timed region of every step): elements + caption ids of a step's screenshots -> the
    fixed-width records the job's one all_gather moves; unpacked again they hold the boxes and, on the captioned icons' rows, the ids."""
    import torch
    import bench
    from omniparser_amd import dist as OD
    B = 2
    recs = torch.zeros(3 * B, OD.REC_W, dtype=torch.int32)
    elems = [[{"bbox": [0.1, 0.2, 0.3, 0.4], "source": "box_ocr_content_ocr"},
              {"bbox": [0.5, 0.5, 0.6, 0.7], "source": "box_yolo_content_yolo"},
              {"bbox": [0.0, 0.1, 0.2, 0.3], "source": "box_yolo_content_ocr"},
              {"bbox": [0.7, 0.7, 0.9, 0.8], "source": "box_yolo_content_yolo"}],
             []]
    ids = [[torch.tensor([2, 0, 11, 12, 2]), torch.tensor([2, 0, 13, 2])], []]
    bench.pack_records(recs, B, torch.device("cpu"), step_id=5, li=1, elems=elems, ids=ids)
    item, boxes, conf, cls, cap = OD.unpack_record(recs[2])
    assert item == 10 and boxes.shape == (4, 4) and torch.allclose(boxes[1], torch.tensor([0.5, 0.5, 0.6, 0.7]))
    assert cap[1, :5].tolist() == [2, 0, 11, 12, 2] and cap[3, :4].tolist() == [2, 0, 13, 2] and int(cap[0].sum()) == 0 and int(cap[2].sum()) == 0
    item2, boxes2, _, _, cap2 = OD.unpack_record(recs[3])
    assert item2 == 11 and boxes2.shape == (0, 4) and cap2.shape[0] == 0
    assert int(recs[:2].abs().sum()) == 0 and int(recs[4:].abs().sum()) == 0          # only this step's rows were written


def test_bench_argument_parser_formats_its_help_and_defaults(monkeypatch, capsys):
    """argparse expands '%' in help strings: a stray one only shows when somebody asks for --help.  Defaults = the driver's contract."""
    import bench
    monkeypatch.setattr("sys.argv", ["bench.py", "--help"])
    with pytest.raises(SystemExit) as e:
        bench.parse_args()
    assert e.value.code == 0 and "--lanes" in capsys.readouterr().out
    monkeypatch.setattr("sys.argv", ["bench.py"])
    a = bench.parse_args()
    assert (a.gpus, a.mode, a.batch, a.lanes, a.pipeline) == (1, "e2e", 8, 2, True)
    assert not any(hasattr(a, k) for k in ("lane_masks", "split_masks", "candidates"))      # round-4: the losing experiments and their switches are gone


def test_build_staleness_is_decided_by_content_not_mtime(tmp_path):
    """omniparser_amd/build.py: an object is stale when the sha256 stamp next to it does not match the content of its sources (+ the
    compiler flags); touching a file — what a snapshot that does not preserve timestamps does to all of them — changes nothing."""
    import os
    from omniparser_amd import build as B
    src, tgt = tmp_path / "a.hip", tmp_path / "a.o"
    src.write_text("int x;\n")
    assert B._stale(tgt, [src])                       # no target
    tgt.write_bytes(b"obj")
    assert B._stale(tgt, [src])                       # no stamp
    digest = B._digest([src])                         # taken BEFORE the (imaginary) compile
    B._mark(tgt, digest)
    assert not B._stale(tgt, [src])
    os.utime(src, (2_000_000_000, 2_000_000_000))     # newer than the target: irrelevant
    os.utime(tgt, (1_000_000_000, 1_000_000_000))
    assert not B._stale(tgt, [src])
    src.write_text("int y;\n")
    assert B._stale(tgt, [src])
    # a source edited WHILE the compiler ran: the stamp records what the object was built from, so the next call rebuilds
    d_before = B._digest([src])
    src.write_text("int z;\n")
    B._mark(tgt, d_before)
    assert B._stale(tgt, [src])
    # the shipped library carries a stamp that matches the tree it was built from
    hdrs = list(B.CSRC.glob("*.h")) + [B.PKG.parent / "include" / "omni_amd.h"]
    assert B.LIB.exists() and not B._stale(B.LIB, B.sources() + hdrs)


def test_conv_tuning_table_is_well_formed_and_reaches_the_op_descriptors(monkeypatch):
    """omniparser_amd/conv_tuning_gfx950.json (tools/conv_autotune.py): every entry is a shape key `MxNxKk<k>s<s>` with a tile code 1..3
    and a split-K count 1..64; a PlanBuilder that carries the table writes the choice into OMNI_OP_CONV i22 / i23 of a matching
    split-f16 conv and leaves every other conv on the launcher's heuristic (0, 0)."""
    import json
    import re
    import torch
    from omniparser_amd.planner import PlanBuilder, View, conv_key
    from omniparser_amd.util import yolov9 as Y
    table = json.loads((ROOT / "omniparser_amd" / "conv_tuning_gfx950.json").read_text())["choices"]
    assert len(table) >= 20
    for k, (tile, splits) in table.items():
        assert re.fullmatch(r"\d+x\d+x\d+k[13]s[12]", k), k
        assert tile in (1, 2, 3) and 1 <= splits <= 64, (k, tile, splits)
    monkeypatch.setattr(Y, "_CONV_TUNING", "unset")
    assert Y.conv_tuning() == {k: tuple(v) for k, v in table.items()}
    monkeypatch.setenv("OMNI_CONV_TUNING", "0")
    monkeypatch.setattr(Y, "_CONV_TUNING", "unset")
    assert Y.conv_tuning() is None
    monkeypatch.setattr(Y, "_CONV_TUNING", "unset")
    # a 1x1 conv of 20 x 20 pixels, 128 -> 128 channels: key 400x128x128k1s1
    pb = PlanBuilder("cpu", L.F32)
    pb.split = True
    pb.conv_tuning = {conv_key(400, 128, 128, 1, 1): (3, 2)}
    x = View(torch.zeros(1, 20, 20, 128), 0, 128)
    o1, o2 = pb.alloc(1, 20, 20, 128), pb.alloc(1, 10, 10, 128)
    w = pb.pack_weight(torch.randn(128, 128, 1, 1))
    pb.conv(x, w, None, o1, 1)
    w3 = pb.pack_weight(torch.randn(128, 128, 3, 3))
    pb.conv(x, w3, None, o2, 3, 2)
    assert pb.ops[0].i[20] == 1 and (pb.ops[0].i[22], pb.ops[0].i[23]) == (3, 2)
    assert (pb.ops[1].i[22], pb.ops[1].i[23]) == (0, 0)


def test_tie_statistics_count_only_decisions_that_can_reach_the_final_list():
    """oracle/detector_ref.py::postprocess counts NMS ties (diagnostics the parity tests select frames by) over the first max_det keeps
    and over victims scoring above the last of them: a tie about a box that can never enter the final list is not a tie of the result.
    Construction: 6 well-separated high-score boxes, then a low-score pair whose IoU sits exactly at the threshold.  With max_det = 4 the
    pair is out of reach (no tie counted, final list = the 4 best); with max_det = 300 the same pair IS a counted near tie.  The kept
    boxes are identical either way — the statistics pass never changes the result."""
    from oracle import detector_ref as D
    far = [[100.0 * k, 0.0, 100.0 * k + 50.0, 50.0] for k in range(6)]
    # IoU([0,0,10,10], [0,0,10,5]) = 0.5 exactly, placed far away from the others
    pair = [[2000.0, 0.0, 2010.0, 10.0], [2000.0, 0.0, 2010.0, 5.0]]
    boxes = torch.tensor(far + pair)
    scores = torch.tensor([0.9, 0.85, 0.8, 0.75, 0.7, 0.65, 0.2, 0.1])
    cls = torch.zeros(8, dtype=torch.long)
    full = D.batched_nms(boxes, scores, cls, 0.5)
    for max_det, want_ties in ((4, 0), (300, 1)):
        keep = full[:max_det]
        stats = {"score_floor": float(scores[keep[-1]]) - 1e-4 if len(full) > max_det else -float("inf")}
        again = D.batched_nms(boxes, scores, cls, 0.5, stats, max_keep=max_det)
        assert again.tolist() == full[:max_det].tolist()                       # an abridged pass keeps the same first max_det boxes
        assert stats.get("near_ties", 0) == want_ties, (max_det, stats)
    # the unabridged call without a floor counts every decision, as rounds 2-4 did
    stats = {}
    assert D.batched_nms(boxes, scores, cls, 0.5, stats).tolist() == full.tolist() and stats["near_ties"] == 1


def test_range_guard_reading_reaches_stats_and_is_fatal_in_strict_mode(monkeypatch):
    """`ScreenParser._check_range` (round 6): what `omni_overflow_count` returns is accumulated and surfaced; OMNI_STRICT_RANGE=1 — the
    setting of this test suite (tests/conftest.py) — turns a non-zero reading into an error."""
    from omniparser_amd.pipeline import ScreenParser
    sp = ScreenParser.__new__(ScreenParser)
    readings = iter([0, 5])
    monkeypatch.setattr(L, "overflow_count", lambda reset=True: next(readings))
    monkeypatch.setenv("OMNI_STRICT_RANGE", "0")
    sp._check_range()
    assert sp.range_overflow_last == 0 and sp.range_overflow_total == 0
    sp._check_range()
    assert sp.range_overflow_last == 5 and sp.range_overflow_total == 5
    monkeypatch.setattr(L, "overflow_count", lambda reset=True: 3)
    monkeypatch.setenv("OMNI_STRICT_RANGE", "1")
    with pytest.raises(L.OmniError, match="range guard"):
        sp._check_range()
    import os
    assert os.environ.get("OMNI_STRICT_RANGE") == "1"


def test_overlay_default_follows_the_models_device(monkeypatch):
    """round 6: with OMNI_OVERLAY unset the annotated image is produced on the device exactly when the models live on a GPU; the
    variable still forces either path (util/utils.py::overlay_on_device)."""
    from omniparser_amd.util import utils as U
    monkeypatch.delenv("OMNI_OVERLAY", raising=False)
    assert U.overlay_on_device(torch.device("cuda", 0)) and U.overlay_on_device("cuda:1")
    assert not U.overlay_on_device(torch.device("cpu")) and not U.overlay_on_device(None) and not U.overlay_on_device(object())
    monkeypatch.setenv("OMNI_OVERLAY", "host")
    assert not U.overlay_on_device(torch.device("cuda", 0))
    monkeypatch.setenv("OMNI_OVERLAY", "device")
    assert U.overlay_on_device(torch.device("cpu"))


def test_scan_provenance_describes_the_tree_it_was_measured_on():
    """bench.py quotes `config.parity_scan` only from a scan whose recorded provenance (conv tuning table, kernel sources) equals the
    tree's (advisor, round 5: the quoted scan predated the tuning table).  The committed round-6 scan must describe THIS tree — a kernel
    edit after the closing session without a new scan fails here, in the CPU suite, instead of silently dropping the citation."""
    import hashlib
    import json
    import re
    scans = sorted((ROOT / "profiles").glob("r*_scan_gpu_vs_oracle.json"), key=lambda p: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", p.name)])
    assert scans and scans[-1].name.startswith("r6_"), [p.name for p in scans]
    sc = json.loads(scans[-1].read_text())
    tune = ROOT / "omniparser_amd" / "conv_tuning_gfx950.json"
    cur = {"conv_tuning_sha16": hashlib.sha256(tune.read_bytes()).hexdigest()[:16], "conv_tuning_applied": True,
           "kernel_sources_sha16": hashlib.sha256(b"".join(p.read_bytes() for p in sorted((ROOT / "omniparser_amd" / "csrc").glob("*.h*")))).hexdigest()[:16]}
    assert {k: sc["provenance"].get(k) for k in cur} == cur
    assert sc["frames"] == 110 and sc["final_boxes_identical"] >= 109 and sc["identical_up_to_exchanges_of_equal_score_neighbours"] >= 109


def test_plan_builder_arena_replay_hands_out_prefixes_and_refuses_what_does_not_fit():
    """PlanBuilder arena mode (round 6: the exact-row twins of florence.py::_CaptionPlans.encode_rows): tensor k of the replaying builder is
    the first bytes of tensor k of the builder that owns the buffers — including tensors carved out of released ones — nothing is
    allocated, released or zeroed; a larger tensor, a longer allocation sequence or a release are errors / no-ops, never a silent overrun."""
    from omniparser_amd.planner import PlanBuilder
    own = PlanBuilder("cpu", L.F32)
    own.reuse = True
    a = own.alloc(4, 6, 6, 8)
    b = own.raw((4 * 100,), torch.float32, zero=False)
    own.release(a)
    c = own.alloc(4, 3, 3, 8)                                # carved out of a's bytes (lifetime reuse)
    z = own.alloc(4, 2, 2, 4, zero=True)
    z.t.fill_(3.0)
    assert c.t.data_ptr() == a.t.data_ptr() and len(own.alloc_log) == 4
    twin = PlanBuilder("cpu", L.F32)
    twin.reuse = True
    twin.arena = iter(own.alloc_log)
    a2 = twin.alloc(3, 6, 6, 8)
    b2 = twin.raw((3 * 100,), torch.float32, zero=False)
    twin.release(a2)                                         # a no-op: lifetimes are the owner's
    c2 = twin.alloc(3, 3, 3, 8)
    z2 = twin.alloc(3, 2, 2, 4, zero=True)
    assert (a2.t.data_ptr(), b2.data_ptr(), c2.t.data_ptr(), z2.t.data_ptr()) == (a.t.data_ptr(), b.data_ptr(), c.t.data_ptr(), z.t.data_ptr())
    assert tuple(a2.t.shape) == (3, 6, 6, 8) and b2.numel() == 300 and not twin._free and not twin.keep
    assert float(z2.t.min()) == 3.0                          # zero-initialised state keeps the owner's content
    a2.t.fill_(5.0)
    assert float(a.t[:3].min()) == 5.0 and float(c.t.view(-1)[0]) == 5.0
    with pytest.raises(RuntimeError, match="more allocations"):
        twin.alloc(1, 1, 1, 4)
    big = PlanBuilder("cpu", L.F32)
    big.arena = iter(own.alloc_log)
    with pytest.raises(RuntimeError, match="does not fit"):
        big.alloc(5, 6, 6, 8)
