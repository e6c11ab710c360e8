"""Summarise a rocprofv3 rocpd (.db) kernel trace as a per-kernel stats table (like --stats CSV)."""
import sqlite3
import sys


def main(path, out=None):
    con = sqlite3.connect(path)
    cur = con.cursor()
    q = """select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name order by 3 desc"""
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows)
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
    for name, n, total, avg, mn, mx in rows:
        lines.append(f'"{name}",{n},{total},{avg:.1f},{mn},{mx},{100.0 * total / tot:.2f}')
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
