"""Are the kernels a GPU session measured still the kernels that ship?  Compiles every csrc/*.hip of a git revision and of the working
tree for gfx950 (same flags as omniparser_amd/build.py) and compares the machine code kernel by kernel (assembly text with local
labels normalised).  Used at the end of round 3: three candidate kernels were added after the last GPU minute — the report shows that
every kernel of the measured default path is byte-for-byte the code the closing GPU session ran.
usage: python tools/isa_diff.py <git-rev> [stem ...]  > profiles/<name>.txt        (no GPU needed)"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from omniparser_amd.build import FLAGS, HIPCC  # noqa: E402

CSRC = ROOT / "omniparser_amd" / "csrc"


def kernels(asm: Path):
    out, cur, body = {}, None, []
    for l in asm.read_text().split("\n"):
        m = re.match(r"^(_Z\w+):", l)
        if m and cur is None:
            cur, body = m.group(1), []
            continue
        if cur is not None:
            if "s_endpgm" in l:
                out[cur] = "\n".join(body)
                cur = None
            else:
                t = l.split(";")[0].rstrip()
                if t.strip() and not t.strip().startswith("."):
                    body.append(re.sub(r"\.LBB\d+_", ".LBB_", t))
    return out


def compile_dir(src_dir: Path, stem: str, work: Path):
    work.mkdir(parents=True, exist_ok=True)
    subprocess.run([HIPCC, *FLAGS, "--save-temps", "-c", str(src_dir / f"{stem}.hip"), "-o", str(work / "o.o")],
                   cwd=work, check=True, capture_output=True)
    return kernels(next(work.glob("*gfx950*.s")))


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return [n.replace("(anonymous namespace)::", "").split("(")[0] for n in r.stdout.strip().split("\n")] if r.returncode == 0 else names


def main():
    rev = sys.argv[1]
    stems = sys.argv[2:] or sorted(p.stem for p in CSRC.glob("*.hip"))
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    dirty = bool(subprocess.run(["git", "status", "--porcelain", "omniparser_amd/csrc", "include"], cwd=ROOT, capture_output=True, text=True).stdout.strip())
    print(f"machine code per kernel: {rev} vs working tree at {head}{' (+ uncommitted changes)' if dirty else ''}; flags {' '.join(FLAGS)}")
    total_same = total_diff = 0
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        old_src = td / "old" / "omniparser_amd" / "csrc"          # same relative layout as the tree (the sources include ../../include/omni_amd.h)
        old_src.mkdir(parents=True)
        (td / "old" / "include").mkdir()
        (td / "old" / "include" / "omni_amd.h").write_bytes(subprocess.run(["git", "show", f"{rev}:include/omni_amd.h"], cwd=ROOT, capture_output=True, check=True).stdout)
        for f in subprocess.run(["git", "ls-tree", "--name-only", rev, "omniparser_amd/csrc/"], cwd=ROOT, capture_output=True, text=True, check=True).stdout.split():
            if f.endswith((".hip", ".h")):
                (old_src / Path(f).name).write_bytes(subprocess.run(["git", "show", f"{rev}:{f}"], cwd=ROOT, capture_output=True, check=True).stdout)
        for stem in stems:
            if not (old_src / f"{stem}.hip").exists():
                print(f"{stem}.hip: new file")
                continue
            old = compile_dir(old_src, stem, td / f"old_{stem}")
            new = compile_dir(CSRC, stem, td / f"new_{stem}")
            same = [k for k in old if k in new and old[k] == new[k]]
            diff = [k for k in old if k in new and old[k] != new[k]]
            gone = [k for k in old if k not in new]
            added = [k for k in new if k not in old]
            # a kernel that only gained a (defaulted) template parameter has a new mangled name: identical if its instructions are
            renamed = {}
            for k in list(gone):
                twin = next((n for n in added if new[n] == old[k]), None)
                if twin is not None:
                    renamed[k] = twin
                    gone.remove(k); added.remove(twin); same.append(k)
            total_same += len(same)
            total_diff += len(diff) + len(gone)
            print(f"{stem}.hip: {len(same)} kernels identical, {len(diff)} changed, {len(gone)} removed, {len(added)} added")
            for a_, b_ in zip(demangle(list(renamed)), demangle(list(renamed.values()))) if renamed else []:
                print(f"    identical under a new name: {a_} -> {b_}")
            for tag, ks in (("changed", diff), ("removed", gone), ("added", added)):
                for n in demangle(ks) if ks else []:
                    print(f"    {tag}: {n}")
    print(f"TOTAL: {total_same} kernels identical, {total_diff} changed or removed")
    return 0


if __name__ == "__main__":
    sys.exit(main())
