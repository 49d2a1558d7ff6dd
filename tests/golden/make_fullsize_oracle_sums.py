#!/usr/bin/env python
"""Runs the CPU ORACLE (oracle/, all host cores) on BASELINE.json's configurations at FULL size and
writes tests/golden/fullsize_oracle_sums.json: one checksum per Tree / FMMTraversalInfo array
(tests/fullsize_sums.py), which tests/test_gpu_fullsize.py compares the HIP path with on the GPU.
TEST INFRASTRUCTURE; runs in the build container (62 GB of host memory, 8 cores), never on the GPU
box, and never imports the product.

    python tests/golden/make_fullsize_oracle_sums.py [--only c2 c3 ...] [--list]

Recipes = SURVEY.md section 8(d), as bench.py::make_workload_numpy draws them (seed 15):
  c1   2D uniform 10^5, mpb 30                       c2   3D uniform 10^7, mpb 64
  c3   3D sphere surface 10^8                        c3c  clustered variant of c3
  c4   10^8 sources + 10^7 targets with radii, stick_out_factor 0.25
  c5w1 one rank's chunk of configs[4]: default_rng(15).random(1.25 * 10^8) x 3 (tree + lists)
  c5w2 chunks 15 and 16 concatenated, 2.5 * 10^8 points: TREE ONLY (its List 2 would not fit the
       reference's int32 CSR starts) -- what worlds["2"] of c5_global_counts.json
       (written by the product) claims, restated by the oracle
  c5r8 eight chunks default_rng(15 + g).random(15 625 000) x 3 = 1.25 * 10^8 points: the N = 8 split
       of configs[4] at the largest size whose SINGLE-tree lists exist in int32; per-list sums that are
       linear over target boxes (fullsize_sums.csr_rows_sum), so eight ranks' sums add up to them
Each configuration runs in its own process (peak memory ~150 B per particle).
"""

from __future__ import annotations

import argparse
import json
import os
import resource
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(HERE, "fullsize_oracle_sums.json")

CONFIGS = {
    "c1": dict(workload="c1", n=10**5, mpb=30),
    "c2": dict(workload="c2", n=10**7, mpb=64),
    "c3": dict(workload="c3", n=10**8, mpb=64),
    "c3c": dict(workload="c3c", n=10**8, mpb=64),
    "c4": dict(workload="c4", n=10**8, mpb=64),
    "c5w1": dict(chunks=1, n_chunk=125_000_000, mpb=64),
    "c5w2": dict(chunks=2, n_chunk=125_000_000, mpb=64, tree_only=True),
    "c5r8": dict(chunks=8, n_chunk=15_625_000, mpb=64, sharded=True),
}


def inputs(cfg):
    if "workload" in cfg:
        from bench import make_workload_numpy
        return make_workload_numpy(cfg["workload"], cfg["n"], 15)
    parts = [[], [], []]
    for g in range(cfg["chunks"]):
        rng = np.random.default_rng(15 + g)
        for ax in range(3):
            parts[ax].append(rng.random(cfg["n_chunk"]))
    return dict(particles=[np.concatenate(p) for p in parts], targets=None, kw={})


def sharded_sums(torch, tree, trav):
    """The sums a sharded build can reproduce rank by rank (distributed/checksum.py for the tree;
    fullsize_sums.csr_rows_sum with global box numbers for the lists)."""
    import fullsize_sums as fs
    import sharded_sums as ss
    nb = int(tree.nboxes)
    gids = torch.arange(nb, dtype=torch.int64)
    lists = ss.single_tree_sums(torch, tree, trav)
    out = {
        "nboxes": nb, "nlevels": int(tree.nlevels),
        "level_start_box_nrs": [int(v) for v in tree.level_start_box_nrs],
        "counts_cumul_checksum": fs.rows_sum(torch, gids, torch.from_numpy(tree.box_source_counts_cumul).to(torch.int64)),
        "user_source_ids_checksum": fs.array_sum(torch, tree.user_source_ids),
        "ntarget_boxes": len(trav.target_boxes),
        "colleagues": lists["colleagues"], "list1": lists["list1"], "list2": lists["list2"],
        "list4": lists["list4"],
        "list3": [lists[f"list3[{lev}]"] for lev in range(int(tree.nlevels))],
        "entries": {"colleagues": len(trav.same_level_non_well_sep_boxes_lists),
                    "list1": len(trav.neighbor_source_boxes_lists),
                    "list2": len(trav.from_sep_siblings_lists),
                    "list4": len(trav.from_sep_bigger_lists),
                    "list3": [int(bl.count) for bl in trav.from_sep_smaller_by_level]},
    }
    return out


def run_one(name):
    import torch
    import fullsize_sums as fs
    from oracle import oracle
    cfg = CONFIGS[name]
    cores = len(os.sched_getaffinity(0))
    oracle.set_variant("omp", cores)
    t0 = time.time()
    w = inputs(cfg)
    t1 = time.time()
    tree = oracle.build_tree(w["particles"], targets=w["targets"], max_particles_in_box=cfg["mpb"], **w["kw"])
    t2 = time.time()
    del w
    entry = {"config": cfg, "oracle": f"oracle/liboracle_omp.so, {cores} threads",
             "tree": fs.tree_sums(torch, tree), "seconds": {"inputs": t1 - t0, "tree": t2 - t1}}
    # the two sums bench.py / tests/test_gpu_c5.py use for sharded builds: the same formulas as
    # boxtree_amd/distributed/checksum.py (tree_checksum = rows_sum over all boxes,
    # particle_order_checksum = array_sum), restated in tests/fullsize_sums.py -- no product import
    entry["tree"]["counts_cumul_checksum"] = fs.rows_sum(
        torch, torch.arange(int(tree.nboxes), dtype=torch.int64),
        torch.from_numpy(tree.box_source_counts_cumul).to(torch.int64))
    entry["tree"]["user_source_ids_checksum"] = fs.array_sum(torch, tree.user_source_ids)
    entry["tree"]["level_start_box_nrs.values"] = [int(v) for v in tree.level_start_box_nrs]
    if not cfg.get("tree_only"):
        t3 = time.time()
        trav = oracle.build_traversal(tree)
        entry["seconds"]["traversal"] = time.time() - t3
        entry["traversal"] = fs.traversal_sums(torch, trav)
        if cfg.get("sharded"):
            entry["sharded"] = sharded_sums(torch, tree, trav)
    entry["seconds"]["sums"] = time.time() - t2 - entry["seconds"].get("traversal", 0.0)
    entry["peak_rss_GB"] = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
    return entry


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="+", default=None)
    ap.add_argument("--child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--list", action="store_true")
    args = ap.parse_args()
    if args.list:
        print(json.dumps(CONFIGS, indent=1))
        return
    if args.child:
        print("RESULT " + json.dumps(run_one(args.child)), flush=True)
        return
    doc = json.load(open(OUT)) if os.path.exists(OUT) else {}
    doc["what"] = ("Checksums (tests/fullsize_sums.py) of the trees and traversals the CPU ORACLE builds from "
                   "BASELINE.json's configurations at full size, written by "
                   "tests/golden/make_fullsize_oracle_sums.py in the build container.  The product never "
                   "wrote any of these numbers; tests/test_gpu_fullsize.py compares the HIP path with them.")
    doc.setdefault("configs", {})
    for name in args.only or list(CONFIGS):
        t0 = time.time()
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name],
                           capture_output=True, text=True)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
        if p.returncode != 0 or not line:
            print(f"[{name}] FAILED rc={p.returncode}\n{p.stderr[-2000:]}", file=sys.stderr)
            continue
        doc["configs"][name] = json.loads(line[0][7:])
        print(f"[{name}] done in {time.time() - t0:.0f} s: nboxes {doc['configs'][name]['tree']['nboxes']}, "
              f"peak {doc['configs'][name]['peak_rss_GB']:.1f} GB", file=sys.stderr, flush=True)
        with open(OUT, "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)
            f.write("\n")


if __name__ == "__main__":
    main()
