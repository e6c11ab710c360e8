"""Build tests/emu/libomni_emu.so: the DEVICE SOURCES of omniparser_amd/csrc/*.hip compiled for the host against the emulation
header tests/emu/fakehip/hip/hip_runtime.h (work-items = threads, waves = 64-thread collectives, MFMA / LDS-DMA / barriers
emulated).  Same C ABI as libomni_amd.so (include/omni_amd.h), host pointers instead of device pointers.  TEST INFRASTRUCTURE."""
import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
CSRC = ROOT / "omniparser_amd" / "csrc"
OUT = HERE / "libomni_emu.so"
OBJ = HERE / "_obj"
CLANG = os.environ.get("OMNI_EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["-x", "c++", "-std=c++20", "-O2", "-mavx2", "-mf16c", "-pthread", "-fPIC", "-ffp-contract=off", "-Wno-unknown-pragmas", "-Wno-unused-value",
         "-Wno-psabi", f"-I{HERE / 'fakehip'}"]


def build(verbose=False) -> Path:
    srcs = sorted(CSRC.glob("*.hip"))
    deps = srcs + sorted(CSRC.glob("*.h")) + [HERE / "fakehip" / "hip" / "hip_runtime.h", ROOT / "include" / "omni_amd.h", Path(__file__)]
    # staleness by CONTENT (sha256 stamp next to the library), like omniparser_amd/build.py: an mtime comparison calls a library fresh
    # that a concurrent build finished AFTER an edit but compiled from the sources BEFORE it
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for d in sorted(map(str, deps)):
        h.update(os.path.basename(d).encode() + b"\0" + Path(d).read_bytes())
    digest, stamp = h.hexdigest(), OUT.with_name(OUT.name + ".sha256")
    if OUT.exists() and stamp.exists() and stamp.read_text().strip() == digest:
        return OUT
    OBJ.mkdir(exist_ok=True)
    procs = []
    for s in srcs:
        o = OBJ / (s.stem + ".o")
        procs.append((s, subprocess.Popen([CLANG, *FLAGS, "-c", str(s), "-o", str(o)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            raise RuntimeError(f"host emulation build failed for {s.name}:\n{out[-6000:]}")
        if verbose:
            print("compiled", s.name)
    subprocess.run([CLANG, "-shared", "-pthread", "-o", str(OUT), *[str(OBJ / (s.stem + ".o")) for s in srcs]], check=True)
    stamp.write_text(digest + "\n")
    return OUT


if __name__ == "__main__":
    print(build(verbose=True))
