"""The device hand-off kernel (omniparser_amd/csrc/glue_ops.hip::glue_kernel) run FROM ITS DEVICE SOURCE on the host: tests/emu/
glue_emu.cpp compiles the .hip file with g++ (256 std::threads per workgroup, std::barrier for __syncthreads) and the GPU check
`gpu_checks.check_glue` — 36 fixtures recorded from the reference's own remove_overlap_new / int_box_area, 40 random trials against
the host twin ScreenParser.glue — runs against it unchanged.  The `-m gpu` twin of this test launches the same source on the MI355X."""
import ctypes
import subprocess
from pathlib import Path

import pytest

HERE = Path(__file__).resolve().parent


@pytest.fixture(scope="module")
def emu():
    src = HERE / "emu" / "glue_emu.cpp"
    out = HERE / "emu" / "libglue_emu.so"
    deps = [src, HERE.parent / "omniparser_amd" / "csrc" / "glue_ops.hip", HERE.parent / "include" / "omni_amd.h"]
    if not out.exists() or out.stat().st_mtime < max(d.stat().st_mtime for d in deps):
        subprocess.run(["g++", "-O1", "-std=c++20", "-pthread", "-shared", "-fPIC", "-Wno-unknown-pragmas", "-o", str(out), str(src)],
                       check=True, cwd=str(src.parent))
    lib = ctypes.CDLL(str(out))
    lib.omni_emu_glue.restype = ctypes.c_char_p
    return lib


def test_device_handoff_source_on_host_matches_reference_fixtures_and_host_twin(emu, monkeypatch):
    import gpu_checks as G
    from omniparser_amd import _lib as L

    def launch(op, stream=None):
        assert op.kind == L.OP_GLUE
        err = emu.omni_emu_glue(ctypes.byref(op))
        assert err is None, err

    monkeypatch.setattr(G, "DEV", "cpu")
    monkeypatch.setattr(G, "_sync", lambda: None)
    monkeypatch.setattr(L, "launch", launch)
    r = G.check_glue()
    assert r["fixture_cases"] >= 36 and r["random_trials"] >= 40


def test_emulated_kernel_rejects_bad_capacity(emu):
    from omniparser_amd import _lib as L
    op = L.make_op(L.OP_GLUE, L.F32, p=[0] * 8, i={0: 300, 1: 10, 2: 1920, 3: 1080, 4: 1, 6: 310})
    assert b"null pointer" in emu.omni_emu_glue(ctypes.byref(op))
