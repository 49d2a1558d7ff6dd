#!/usr/bin/env python
"""HBM traffic per launch of the largest kernels of a workload, from two rocprofv3 PMC
passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) over bench.py, corrected as
MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts half of the bytes read;
calibrated here on the streaming kernels of the same run whose byte counts are known),
next to each kernel's algorithmic bytes.

    python tools/pmc_traffic.py WORKLOAD N FETCH.csv WRITE.csv KERNEL_STATS.csv OUT.json
"""
import csv
import json
import sys


def load(path):
    out = {}
    for row in csv.DictReader(open(path)):
        out[row["Kernel"]] = (int(row["Dispatches"]), float(row["Avg"]), float(row["Max"]))
    return out


def load_stats(path):
    out = {}
    try:
        for row in csv.DictReader(open(path)):
            name = row.get("Name") or row.get("KernelName") or row.get("Kernel")
            avg = row.get("AverageNs") or row.get("Average") or row.get("avg_ns")
            if name and avg:
                out[name] = float(avg) * 1e-6
    except OSError:
        pass
    return out


# per-kernel algorithmic bytes per particle (N) of ONE launch of the biggest instance, and why
SPECS = {
    "c3": [
        ("gather_packed_kernel", 60, "4 (id) + 32 (one padded record, random) + 24 (coordinates out)"),
        ("keygen_kernel", 64, "24 (coordinates) + 8 (key) + 32 (padded record out)"),
        ("onesweep_keys_kernel", 16, "8 read + 8 written: one packed word per particle"),
        ("segment_sort_wave_kernel", 12, "8 (packed word) + 4 (id out); box arrays apart"),
        ("scatter_inverse_kernel", 12, "8 (id, position) + 4 (scatter)"),
        ("bbox_axes_kernel", 24, "coordinates"),
        ("box_extent_kernel", 24, "coordinates of the leaves (deepest levels), box arrays apart"),
        ("split_level_kernel", None, "binary searches in the keys + box arrays: no per-particle figure"),
        ("walk13_v2_kernel", None, "4 bytes per list entry written; reads are box records"),
        ("coll_rows_v3_kernel", None, "4 bytes per list entry written"),
        ("level_tables_kernel", None, "4 bytes per row entry written"),
    ],
}
# the N > 1 code path on one rank (bench.py --force-dist): the exchange's own kernels
SPECS["c3dist"] = SPECS["c3"] + [
    ("morton_cells_lds_kernel", 28, "24 (coordinates) + 4 (cell out); the histogram stays in LDS"),
    ("pp_count_kernel", 5, "4 (cell) + 1 (owner byte out)"),
    ("pp_scatter_kernel", 49, "1 (owner) + 24 (coordinates) + 24 (record out)"),
    ("let_link_kernel", None, "per box: path lookups in the level above, parent / child / centre out"),
]
SPECS["c4"] = SPECS["c3"]
SPECS["c2"] = SPECS["c3"]
SPECS["c5"] = SPECS["c3"] + [
    ("rows_to_csr_v2_kernel", None, "4 bytes per list-1 entry read and written"),
    ("l1_finalize32_kernel", None, "4 bytes per list-1 entry read and written, one rank lookup each"),
    ("l3_scatter_v2_kernel", None, "4 bytes per list-3 entry read and written"),
    ("list4_lattice_kernel", None, "4 bytes per list-4 entry written"),
]


def main(workload, n, fetch_csv, write_csv, stats_csv, out_path):
    n = int(n)
    fetch, write = load(fetch_csv), load(write_csv)
    # calibration: keygen reads 24 N (coordinates) and writes 40 N, both streaming
    cal = {}
    for name, (cnt, avg, mx) in fetch.items():
        if name.startswith("keygen_kernel"):
            cal["fetch_correction"] = 24.0 * n / (mx * 1024.0)
    for name, (cnt, avg, mx) in write.items():
        if name.startswith("keygen_kernel"):
            cal["write_correction"] = 40.0 * n / (mx * 1024.0)
    fc = 2.0                      # MI355X_MICROARCH.md, HBM section
    wc = 1.0
    kernels = []
    for key, per_n, why in SPECS[workload]:
        f = [(k, v) for k, v in fetch.items() if k.startswith(key) or ("::" + key) in k]
        w = [(k, v) for k, v in write.items() if k.startswith(key) or ("::" + key) in k]
        if not f and not w:
            continue
        # the biggest instance (template variant / level) of the kernel
        fk, fv = max(f, key=lambda kv: kv[1][2]) if f else (None, (0, 0.0, 0.0))
        wk, wv = max(w, key=lambda kv: kv[1][2]) if w else (None, (0, 0.0, 0.0))
        rd = fv[2] * 1024.0 * fc
        wr = wv[2] * 1024.0 * wc
        ent = {"kernel": fk or wk, "hbm_read_bytes_per_launch": int(rd),
               "hbm_write_bytes_per_launch": int(wr), "traffic_bytes_per_launch": int(rd + wr),
               "FETCH_SIZE_KB_max_raw": fv[2], "WRITE_SIZE_KB_max_raw": wv[2],
               "algorithmic": why}
        if per_n is not None:
            ent["algorithmic_bytes_per_launch"] = int(per_n * n)
            ent["traffic_over_algorithmic"] = round((rd + wr) / (per_n * n), 3)
        kernels.append(ent)
    out = {
        "command": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --workload {workload} "
                   "--steps 2 --warmup 1 --cpu-sample 0   (and a second pass with --pmc WRITE_SIZE); "
                   "tools/r3_session.sh; the largest launch of each kernel (max over dispatches)",
        "workload": workload, "n_particles": n,
        "corrections": {"fetch": fc, "write": wc,
                        "note": "gfx950 / ROCm 7.2: FETCH_SIZE reports half of the bytes read "
                                "(MI355X_MICROARCH.md); checked in this run on keygen_kernel, which "
                                "streams 24 N bytes in and 40 N out", "measured_on_keygen": cal},
        "kernels": kernels,
    }
    json.dump(out, open(out_path, "w"), indent=1)
    for k in kernels:
        print(f"{k['kernel'][:60]:60s} {k['traffic_bytes_per_launch'] / 1e9:8.3f} GB "
              f"x{k.get('traffic_over_algorithmic', '-')}")


if __name__ == "__main__":
    main(*sys.argv[1:])
