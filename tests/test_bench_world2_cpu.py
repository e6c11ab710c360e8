"""The N > 1 path of bench.py before the driver runs it: two processes, gloo, the product's device path on the host emulation of
its own kernels (tests/emu) at debug sizes (quarter-width detector, 640x480 frames, 320x320 network input, detect mode — the
distributed skeleton is the same lines for both modes)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_main_world2_gloo_on_the_emulation():
    from tools.make_weights import ensure_blob
    ensure_blob(seed=0, nc=1, width=0.25)                       # the stand-in checkpoint both ranks load (rank 0 would build it otherwise)
    port = _free_port()
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--mode", "detect", "--batch", "1", "--width", "0.25", "--frame", "640x480",
            "--imgsz", "320", "--no-extra", "--no-cpu-baseline"]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMNI_DIST_BACKEND="gloo", OMNI_VERIFY_IMPORT="0", OMNI_BENCH_WATCHDOG="900")
        env.pop("OMNI_EMU", None)
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / "emu" / "bench_emulated.py"), *args], env=env, cwd=str(ROOT),
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=1500) for p in procs]
    for rank, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{se[-3000:]}"
    assert not any(l.startswith("{") for l in outs[1][0].splitlines()), "only rank 0 prints the JSON line"     # (gloo itself prints a connection note)
    lines = [l for l in outs[0][0].strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, outs[0][0][-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["unit"] == "screenshots/s"
    # whole-job throughput: 2 steps x 1 screenshot x 2 ranks over the max-over-ranks time
    assert abs(d["value"] - 4 / (d["ms_per_step"] * 2 / 1000.0)) < 1e-3 * d["value"] + 1e-3
    assert "replicas x2" in d["config"]["parallelism"] and d["config"]["debug"]["frame"] == "640x480"
    assert "cpu_baseline" not in d or d["cpu_baseline"] is None or "skipped" in json.dumps(d["cpu_baseline"])
    # every screenshot of the job arrived in the gathered records: mean kept boxes over all 4 items is a real count
    assert d["config"]["mean_elements_per_screenshot"] > 5


def _check_world2_line(d):
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["unit"] == "screenshots/s"
    assert abs(d["value"] - 4 / (d["ms_per_step"] * 2 / 1000.0)) < 1e-3 * d["value"] + 1e-3
    assert "replicas x2" in d["config"]["parallelism"]
    assert d["config"]["mean_elements_per_screenshot"] > 5


def test_bench_self_launch_world2_gloo_on_the_emulation():
    """`python bench.py --gpus 2` with NO launcher environment (how a driver that does not use torchrun starts it): the script spawns its
    two ranks itself (omniparser_amd.dist.self_launch), rank 0's JSON line is the parent's stdout, and the line equals in structure what
    the torchrun-style invocation above prints."""
    from tools.make_weights import ensure_blob
    ensure_blob(seed=0, nc=1, width=0.25)
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--mode", "detect", "--batch", "1", "--width", "0.25", "--frame", "640x480",
            "--imgsz", "320", "--no-extra", "--no-cpu-baseline"]
    env = dict(os.environ, OMNI_DIST_BACKEND="gloo", OMNI_VERIFY_IMPORT="0", OMNI_BENCH_WATCHDOG="900")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "OMNI_EMU"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "emu" / "bench_emulated.py"), *args], env=env, cwd=str(ROOT),
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    _check_world2_line(json.loads(lines[0]))


def test_self_launch_propagates_a_failing_rank(tmp_path):
    """one rank exits non-zero while the other would wait forever: the launcher terminates the survivor (its exact PID) and returns the
    failing status instead of hanging."""
    script = tmp_path / "ranks.py"
    script.write_text(
        "import os, sys, time\n"
        "assert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
        "assert os.environ['LOCAL_RANK'] == os.environ['RANK']\n"
        "if os.environ['RANK'] == '1':\n    sys.exit(7)\n"
        "time.sleep(600)\n")
    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "from omniparser_amd.dist import self_launch\n"
            "t = time.time(); rc = self_launch(2, argv=[%r]); print(rc, time.time() - t < 60)\n") % (str(ROOT), str(script))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split() == ["7", "True"], r.stdout
