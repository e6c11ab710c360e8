"""Instruction ORDER of a gfx950 kernel's loops (MFMA / ds_read / LDS-DMA / waits / barriers), from the compiler's assembly:
`python tools/isa_seq.py gemm_dma 'ILi256ELi256ELi2ELi4ELi2ELi0ELb0ELb0ELi2E'` prints every loop of the kernels whose mangled name
contains the filter.  Complements tools/isa_report.py (counts): shows whether sched_group_barrier interleavings were honoured."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from omniparser_amd.build import FLAGS, HIPCC  # noqa: E402

SHORT = [(r"v_mfma\S*", "MFMA"), (r"ds_read_b128", "DSR"), (r"ds_read\S*", "dsr"), (r"ds_write\S*", "DSW"), (r"buffer_load_dwordx4.*lds", "DMA"),
         (r"buffer_load\S*", "BLD"), (r"global_load\S*", "GLD"), (r"global_store\S*", "GST"), (r"buffer_store\S*", "BST")]


def main():
    stem, flt = sys.argv[1], sys.argv[2]
    with tempfile.TemporaryDirectory() as d:
        subprocess.run([HIPCC, *FLAGS, "--save-temps", "-c", str(ROOT / "omniparser_amd" / "csrc" / f"{stem}.hip"), "-o", "o.o"], cwd=d, check=True,
                       capture_output=True)
        txt = next(Path(d).glob("*gfx950*.s")).read_text().split("\n")
    starts = [i for i, l in enumerate(txt) if re.match(r"^_Z\w+:", l)]
    for k, i in enumerate(starts):
        name = txt[i].split(":")[0]
        if flt not in name:
            continue
        end = starts[k + 1] if k + 1 < len(starts) else len(txt)
        body = txt[i:end]
        labels = {m.group(1): j for j, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
        print("==", name)
        for j, l in enumerate(body):
            m = re.search(r"s_cbranch\w*\s+(\.LBB\d+_\d+)", l)
            if not (m and m.group(1) in labels and labels[m.group(1)] < j):
                continue
            seq = []
            for l2 in body[labels[m.group(1)]:j]:
                l2 = l2.strip()
                if l2.startswith("s_waitcnt"):
                    seq.append("[" + l2.split(None, 1)[1].replace("lgkmcnt", "lgkm").replace("vmcnt", "vm") + "]")
                elif l2.startswith("s_barrier"):
                    seq.append("|BARRIER|")
                else:
                    for pat, s in SHORT:
                        if re.match(pat, l2):
                            seq.append(s)
                            break
            if "MFMA" in seq:
                print(f"-- loop {m.group(1)} ({j - labels[m.group(1)]} lines):")
                print(" ".join(seq))


if __name__ == "__main__":
    main()
