#!/usr/bin/env python
"""profiles/*_pmc_onesweep_keys.json from the two PMC passes over tools/sort_bench.py keys
(tools/r3_final.sh): python tools/pmc_sort_json.py FETCH.csv WRITE.csv N OUT.json"""
import csv
import json
import sys


def row(path, key):
    for r in csv.DictReader(open(path)):
        if key in r["Kernel"]:
            return r["Kernel"], float(r["Avg"])
    raise KeyError(key)


fetch_csv, write_csv, n, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
name, f = row(fetch_csv, "onesweep_keys_kernel")
_, w = row(write_csv, "onesweep_keys_kernel")
_, hf = row(fetch_csv, "keys_hist_kernel")
rd, wr = f * 1024 * 2.0, w * 1024 * 1.0
alg = 16 * n
json.dump({
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/sort_bench.py keys %d 2 36   "
               "(and a second, separate pass with --pmc WRITE_SIZE); tools/r3_final.sh" % n,
    "kernel": name + "  (keys-only digit pass: 9-bit digits, 16384-key tiles, 32-bit look-back words)",
    "n_pairs": n,
    "algorithmic_bytes_per_launch": alg,
    "FETCH_SIZE_KB_avg_raw": f, "WRITE_SIZE_KB_avg_raw": w,
    "calibration": {
        "note": "gfx950/ROCm 7.2: FETCH_SIZE reports 1/2 of streamed read bytes (MI355X_MICROARCH.md, "
                "HBM section).  Verified in the same run: keys_hist_kernel reads 8*n = %.1f KB and "
                "reports %.1f KB (x%.3f)." % (8 * n / 1024, hf, 8 * n / 1024 / hf),
        "fetch_correction": 2.0, "write_correction": 1.0},
    "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr),
    "traffic_bytes_per_launch": int(rd + wr),
    "traffic_over_algorithmic": round((rd + wr) / alg, 3),
    "reading": "reads exceed 8*n by the look-back polls (512 words per tile row), writes by the "
               "partial 64-byte lines at the ends of the ~32-key digit runs of a 16384-key tile",
}, open(out, "w"), indent=1)
print(open(out).read())
