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


def _mark(target: Path, digest: str) -> None:
    """`digest` is the one computed BEFORE the compile started: a source edited while hipcc runs leaves a stamp that does not match
    it, so the next call rebuilds instead of trusting an object made from the older content."""
    tmp = _stamp(target).with_name(_stamp(target).name + f".tmp{os.getpid()}")
    tmp.write_text(digest + "\n")
    os.replace(tmp, _stamp(target))


class _BuildLock:
    """One builder at a time per tree: N ranks of `bench.py --gpus N` (or pytest-xdist workers) all call `ensure_built()`; without
    the lock they would compile and link the same `_obj/*.o`, `libomni_amd.so` and stamps concurrently.  flock on a file in `_obj/`;
    the waiters find everything fresh when they get the lock."""

    def __enter__(self):
        import fcntl
        OBJ.mkdir(exist_ok=True)
        self.f = open(OBJ / ".build.lock", "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()


def build_lib(force: bool = False, verbose: bool = True) -> Path:
    with _BuildLock():
        return _build_lib_locked(force, verbose)


def _build_lib_locked(force: bool, verbose: bool) -> Path:
    srcs = sources()
    hdrs = list(CSRC.glob("*.h")) + [PKG.parent / "include" / "omni_amd.h"]
    OBJ.mkdir(exist_ok=True)
    jobs = []
    for src in srcs:
        obj = OBJ / (src.stem + ".o")
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj, _digest([src] + hdrs)))
    lib_digest = _digest(srcs + hdrs)

    def compile_one(job):
        src, obj, digest = job
        tmp = obj.with_name(obj.name + f".tmp{os.getpid()}")
        cmd = [HIPCC, *FLAGS, "-c", str(src), "-o", str(tmp)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            tmp.unlink(missing_ok=True)
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stderr}")
        os.replace(tmp, obj)
        _mark(obj, digest)
        return src.name

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for name in ex.map(compile_one, jobs):
                if verbose:
                    print(f"[omniparser_amd.build] compiled {name}", file=sys.stderr)
    objs = [OBJ / (s.stem + ".o") for s in srcs]
    if force or jobs or _stale(LIB, srcs + hdrs):
        tmp = LIB.with_name(LIB.name + f".tmp{os.getpid()}")
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(tmp), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            tmp.unlink(missing_ok=True)
            raise RuntimeError(f"link failed:\n{r.stderr}")
        os.replace(tmp, LIB)                       # a process that already mapped the old file keeps its inode
        _mark(LIB, lib_digest)
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
