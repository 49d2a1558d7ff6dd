#!/usr/bin/env python
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database -> CSV."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
        "from counters_collection group by kernel_name, counter_name "
        "order by sum(value) desc").fetchall()
    lines = ["Kernel,Counter,Dispatches,Avg,Min,Max"]
    for name, ctr, n, avg, mn, mx in rows:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-80:]
        lines.append(f"\"{short}\",{ctr},{n},{avg:.3f},{mn:.3f},{mx:.3f}")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:])
