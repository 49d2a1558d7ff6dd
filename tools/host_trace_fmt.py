#!/usr/bin/env python
"""Format the last step of a BT_HOST_TRACE log: time since the step began and since the
previous stamp."""
import sys

lines = [ln.split() for ln in open(sys.argv[1]) if ln.startswith("[bt-host]") or ln.startswith("[py]")]
last = max(i for i, ln in enumerate(lines) if ln[0] == "[py]" and ln[1] == "step")
t0 = prev = float(lines[last][-1])
for ln in lines[last:]:
    t = float(ln[-1])
    print(f"{ln[0]:>9s} {' '.join(ln[1:-1]):<16s} {t - t0:9.1f} us  (+{t - prev:7.1f})")
    prev = t
