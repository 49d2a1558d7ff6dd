#!/usr/bin/env python
"""One line per bench JSON: ms/step, particles/s, roofline fraction, stage sums.  usage: bench_brief.py FILE..."""
import json
import sys

for path in sys.argv[1:]:
    try:
        line = json.loads(open(path).read().strip().splitlines()[-1])
    except (OSError, ValueError, IndexError) as e:
        print(f"{path}: unreadable ({e})")
        continue
    st = line.get("stages_ms", {})
    tree = sum(v for k, v in st.items() if not k.startswith("trav:") and not k.startswith("x:"))
    trav = sum(v for k, v in st.items() if k.startswith("trav:"))
    extra = {k: v for k, v in line["config"].items() if "loopback" in k or "c5_check" in k}
    xs = {k: round(v, 3) for k, v in st.items() if k.startswith("x:")}
    print(f"{path}: {line['ms_per_step']:.2f} ms  {line['value']:.3g}/s  frac {line['roofline']['frac']:.3f}  "
          f"tree {tree:.2f} trav {trav:.2f} {xs} {extra}")
