#!/usr/bin/env python3
"""Turns a rocprofv3 rocpd sqlite database (default output format of ROCm 7.2's
`rocprofv3 --kernel-trace --stats`) into the per-kernel summary table that
`--stats` prints for CSV output: calls, total/avg/min/max duration, % of GPU time.
usage: summarize_rocpd.py results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
    ksym = [r[1] for r in c.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "display_name" if "display_name" in ksym else ("kernel_name" if "kernel_name" in ksym else ksym[-1])
    q = ("select s.%s, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
         "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
         "group by s.%s order by 3 desc" % (name_col, name_col))
    rows = list(c.execute(q))
    tot = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    for r in rows:
        nm = r[0]
        if len(nm) > 90:
            nm = nm[:87] + "..."
        lines.append("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.2f |" % (nm, r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot))
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "a").write(out)
    print(out)


if __name__ == "__main__":
    main()
