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


def _digest(paths) -> str:
    """sha256 over the compiler flags and the CONTENT of `paths` — staleness is decided by content, never by mtime: whether a
    snapshot of the tree preserves timestamps must not decide if the GPU box runs the shipped library or a fresh compile."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for p in sorted(map(str, paths)):
        h.update(os.path.basename(p).encode() + b"\0")
        h.update(Path(p).read_bytes())
    return h.hexdigest()


def _stamp(target: Path) -> Path:
    return target.with_name(target.name + ".sha256")


def _stale(target: Path, deps) -> bool:
    st = _stamp(target)
    return not target.exists() or not st.exists() or st.read_text().strip() != _digest(deps)


def _mark(target: Path, deps) -> None:
    _stamp(target).write_text(_digest(deps) + "\n")


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
        _mark(obj, [src] + hdrs)
        return src.name

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for name in ex.map(compile_one, jobs):
                if verbose:
                    print(f"[omniparser_amd.build] compiled {name}", file=sys.stderr)
    objs = [OBJ / (s.stem + ".o") for s in srcs]
    if force or jobs or _stale(LIB, srcs + hdrs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        _mark(LIB, srcs + hdrs)
        if verbose:
            print(f"[omniparser_amd.build] linked {LIB}", file=sys.stderr)
    return LIB


def ensure_built() -> Path:
    """Return the library path; with hipcc present the library is rebuilt when the CONTENT of any source differs from what the
    shipped objects were compiled from (sha256 stamps next to the objects and the library, which travel with them) and a failing
    compile RAISES (a stale .so must never run in place of the sources the tests were written against).  Without hipcc a library
    whose stamp does not match the sources is refused as well."""
    if os.path.exists(HIPCC):
        return build_lib(verbose=False)
    if not LIB.exists():
        raise RuntimeError(f"{LIB} is missing and hipcc is unavailable: run python -m omniparser_amd.build")
    hdrs = list(CSRC.glob("*.h")) + [PKG.parent / "include" / "omni_amd.h"]
    if _stale(LIB, sources() + hdrs):
        raise RuntimeError(f"{LIB} was not built from the sources in this tree and hipcc is unavailable")
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(LIB)
