#!/usr/bin/env python
"""Per-step summary of a rocprofv3 kernel-stats CSV: python tools/prof_summary.py CSV [nsteps] [top]
(nsteps 0 or omitted: the number of keygen_kernel launches, one per build; workload
generation and the copy-rate probe of bench.py are in the file too, outside the steps)"""
import csv
import sys


def main(path, nsteps=0, top=40):
    rows = list(csv.DictReader(open(path)))
    if nsteps <= 0:
        nsteps = max([int(r["Calls"]) for r in rows if "keygen_kernel" in r["Name"]] or [6])
    rows = [r for r in rows if "at::native" not in r["Name"]]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"{path}: {tot / nsteps / 1e6:.3f} ms/step kernel time, "
          f"{sum(int(r['Calls']) for r in rows) / nsteps:.0f} launches/step")
    for r in rows[:top]:
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        print(f"  {name[:72]:72s} calls={int(r['Calls']) / nsteps:>5.1f} "
              f"ms/step={float(r['TotalDurationNs']) / nsteps / 1e6:8.3f} max={float(r['MaxNs']) / 1e6:.3f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0,
         int(sys.argv[3]) if len(sys.argv) > 3 else 40)
