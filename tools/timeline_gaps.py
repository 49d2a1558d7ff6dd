#!/usr/bin/env python
"""Idle time between kernels of the last bench step in a rocprofv3 rocpd database.

usage: timeline_gaps.py DB [first-kernel-substring] [min-gap-us] [--kernels] [--all]

A step starts at the first launch whose name contains the given substring (default
"bbox_") with no such launch among the 20 before it; the last complete step is printed as a
timeline: every gap of at least min-gap-us (default 4) with the kernels either side,
and the totals (span, busy, idle, launches)."""
import sqlite3
import sys


def main(path, first="bbox_", min_gap_us="4"):
    min_gap = float(min_gap_us) * 1e3
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    # (separate targets have a bounding-box pass of their own a few launches later)
    starts = [i for i, r in enumerate(rows)
              if first in r[0] and not any(first in q[0] for q in rows[max(0, i - 20):i])]
    if len(starts) < 2:
        print("fewer than two steps found")
        return
    lo, hi = starts[-2], starts[-1]
    step = rows[lo:hi]
    t0 = step[0][1]
    busy = sum(e - s for _, s, e in step)
    span = step[-1][2] - t0
    print(f"launches {len(step)}  span {span / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms  "
          f"idle {(span - busy) / 1e6:.3f} ms")
    gaps = []
    for (n0, s0, e0), (n1, s1, e1) in zip(step[:-1], step[1:]):
        g = s1 - e0
        gaps.append(g)
        if g >= min_gap:
            print(f"  t={(e0 - t0) / 1e3:9.1f} us  gap {g / 1e3:7.1f} us   "
                  f"{n0.split('(')[0][-40:]:>40} -> {n1.split('(')[0][-40:]}")
    if "--kernels" in sys.argv:
        print("kernels of the step in launch order (>= 20 us):")
        for n_, s_, e_ in step:
            if e_ - s_ >= 20e3:
                print(f"  t={(s_ - t0) / 1e3:9.1f} us  {(e_ - s_) / 1e3:8.1f} us  "
                      f"{n_.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:70]}")
    if "--all" in sys.argv:
        print("every launch of the step:")
        for k, (n_, s_, e_) in enumerate(step):
            print(f"  {k:3d} t={(s_ - t0) / 1e3:9.1f} us  {(e_ - s_) / 1e3:8.1f} us  "
                  f"{n_.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:70]}")
    small = sum(g for g in gaps if 0 < g < min_gap)
    print(f"gaps below {min_gap / 1e3:.0f} us: {small / 1e6:.3f} ms in "
          f"{sum(1 for g in gaps if 0 < g < min_gap)} places; "
          f"overlapping launches: {sum(1 for g in gaps if g <= 0)}")


if __name__ == "__main__":
    main(*[a for a in sys.argv[1:] if not a.startswith("--")])
