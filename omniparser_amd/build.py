"""Build libomni_amd.so (gfx950) in-tree with hipcc.

The shared object is git-ignored but travels to the GPU box with the gpurun snapshot.
`python -m omniparser_amd.build` rebuilds it; `ensure_built()` is what the loader calls.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libomni_amd.so"
OBJ = PKG / "csrc" / "_obj"

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-ffp-contract=off",                             # parity: no silent FMA contraction
    "-fhip-fp32-correctly-rounded-divide-sqrt",      # IEEE f32 divide (torch CPU semantics)
    "-Wno-unused-result",
]


def sources():
    return sorted(CSRC.glob("*.hip"))


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build_lib(force: bool = False, verbose: bool = True) -> Path:
    srcs = sources()
    hdrs = list(CSRC.glob("*.h")) + [PKG.parent / "include" / "omni_amd.h"]
    OBJ.mkdir(exist_ok=True)
    jobs = []
    for src in srcs:
        obj = OBJ / (src.stem + ".o")
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [HIPCC, *FLAGS, "-c", str(src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stderr}")
        return src.name

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for name in ex.map(compile_one, jobs):
                if verbose:
                    print(f"[omniparser_amd.build] compiled {name}", file=sys.stderr)
    objs = [OBJ / (s.stem + ".o") for s in srcs]
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print(f"[omniparser_amd.build] linked {LIB}", file=sys.stderr)
    return LIB


def ensure_built() -> Path:
    """Return the library path; with hipcc present the library is rebuilt when any source is newer and a failing compile
    RAISES (a stale .so must never run in place of the sources the tests were written against)."""
    if os.path.exists(HIPCC):
        return build_lib(verbose=False)
    if not LIB.exists():
        raise RuntimeError(f"{LIB} is missing and hipcc is unavailable: run python -m omniparser_amd.build")
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(LIB)
