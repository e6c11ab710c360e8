"""Static report of the gfx950 code objects: per kernel VGPR/AGPR/SGPR, scratch, LDS, waves/SIMD allowed by
registers, and the instruction mix of its hottest loop (MFMA / ds_read / ds_write / global_load / VALU /
s_waitcnt / s_barrier).  Needs only hipcc (no GPU): `python tools/isa_report.py conv_igemm [filter]`.
Used to sanity-check kernels before they get GPU time (spills, occupancy cliffs, loop balance)."""
import re
import subprocess
import sys
import tempfile
from collections import Counter
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from omniparser_amd.build import FLAGS, HIPCC  # noqa: E402


def compile_to_asm(stem: str, workdir: Path) -> Path:
    src = ROOT / "omniparser_amd" / "csrc" / f"{stem}.hip"
    subprocess.run([HIPCC, *FLAGS, "--save-temps", "-c", str(src), "-o", str(workdir / "o.o")], cwd=workdir, check=True,
                   capture_output=True)
    return next(workdir.glob("*gfx950*.s"))


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.strip().split("\n") if r.returncode == 0 else names


def waves_per_simd(total_regs: int) -> int:
    alloc = -(-total_regs // 8) * 8
    return max(min(8, 512 // max(alloc, 1)), 0)


def loop_mix(body):
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    # the hot loop = an INNERMOST loop holding MFMAs (no other MFMA loop nested inside it); the one with the most MFMAs if several
    def n_mfma(a, b):
        return sum(1 for l in body[a:b + 1] if l.strip().startswith("v_mfma"))
    with_mfma = [(a, b) for a, b in loops if n_mfma(a, b) > 0]
    inner = [(a, b) for a, b in with_mfma if not any((c, d) != (a, b) and a <= c and d <= b for c, d in with_mfma)]
    best = max(inner, key=lambda ab: n_mfma(*ab)) if inner else (max(loops, key=lambda ab: ab[1] - ab[0]) if loops else None)
    if best is None:
        return None
    c = Counter()
    for l in body[best[0]:best[1] + 1]:
        t = l.split(";")[0].strip().split()
        if not t or t[0].endswith(":") or t[0].startswith("."):
            continue
        op = t[0]
        for key in ("v_mfma", "ds_read", "ds_write", "global_load", "buffer_load", "global_store", "s_waitcnt", "s_barrier",
                    "s_cbranch", "v_cvt", "v_", "s_"):
            if op.startswith(key):
                c[key] += 1
                break
    return c


def report(stem: str, flt: str = ""):
    with tempfile.TemporaryDirectory() as td:
        asm = compile_to_asm(stem, Path(td)).read_text().split("\n")
    starts = [i for i, l in enumerate(asm) if re.match(r"^_Z\w+:", l)]
    rows = []
    for st in starts:
        name = asm[st].split(":")[0]
        en = next((i for i in range(st, len(asm)) if ".end_amdhsa_kernel" in asm[i]), None)
        if en is None:          # a _Z label that is not a kernel entry (device data, e.g. the range-guard counter's section)
            continue
        body = asm[st:en]
        meta = {}
        for l in body:
            m = re.match(r"\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|accum_offset|private_segment_fixed_size|group_segment_fixed_size)\s+(\d+)", l)
            if m:
                meta[m.group(1)] = int(m.group(2))
        rows.append((name, meta, loop_mix(body)))
    names = demangle([r[0] for r in rows])
    out = []
    for (name, meta, mix), dn in zip(rows, names):
        short = re.sub(r"\(anonymous namespace\)::", "", dn).split("(")[0].replace("void ", "")
        if flt and flt not in short:
            continue
        regs = meta.get("next_free_vgpr", 0)
        line = (f"{short:58s} regs {regs:4d} (accum_offset {meta.get('accum_offset', 0):3d})  waves/SIMD {waves_per_simd(regs)}  "
                f"scratch {meta.get('private_segment_fixed_size', 0):4d} B  LDS {meta.get('group_segment_fixed_size', 0):6d} B")
        if mix and mix.get("v_mfma"):
            line += ("  | loop: mfma %d ds_read %d ds_write %d gload %d valu %d(cvt %d) waitcnt %d barrier %d branch %d"
                     % (mix["v_mfma"], mix["ds_read"], mix["ds_write"], mix["global_load"] + mix["buffer_load"], mix["v_"] + mix["v_cvt"],
                        mix["v_cvt"], mix["s_waitcnt"], mix["s_barrier"], mix["s_cbranch"]))
        out.append(line)
    return out


if __name__ == "__main__":
    stem = sys.argv[1] if len(sys.argv) > 1 else "conv_igemm"
    for l in report(stem, sys.argv[2] if len(sys.argv) > 2 else ""):
        print(l)
