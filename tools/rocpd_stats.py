#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (the default output of
`rocprofv3 --kernel-trace --stats` on ROCm 7.2) as a per-kernel stats table
(same columns as the classic kernel_stats.csv) -> stdout / CSV."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
    for name, calls, tot, avg, mn, mx in rows:
        lines.append(f"\"{name}\",{calls},{tot},{avg:.1f},{100.0 * tot / total:.2f},{mn},{mx}")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:])
