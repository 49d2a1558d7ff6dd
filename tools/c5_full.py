#!/usr/bin/env python
"""BASELINE configs[4] at FULL size on ONE GPU: the tree of the first W chunks of the c5
recipe (chunk g = 3 x default_rng(15 + g).random(1.25e8), W in {1, 2, 4, 8}; W = 8 is the
10^9-point tree) -- tree only, its List 2 exceeds the reference's int32 CSR --, checked on
the device with the reference's tree assertions (tests/device_invariants.py), and recorded
as tests/golden/c5_global_counts.json: box and level counts and a checksum of
box_source_counts_cumul that is linear in the counts (boxtree_amd/distributed/checksum.py),
so that `bench.py --gpus W --workload c5` -- W ranks, each holding one chunk -- can compare
the global numbering it arrives at with the tree one GPU builds from all chunks.

    python tools/c5_full.py --worlds 1 2 4 8 [--n 125000000] [--write]
"""

from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden", "c5_global_counts.json")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--n", type=int, default=125_000_000, help="points per chunk")
    ap.add_argument("--mpb", type=int, default=64)
    ap.add_argument("--write", action="store_true", help="write tests/golden/c5_global_counts.json")
    ap.add_argument("--out", default=None, help="also write the report here (json)")
    ap.add_argument("--no-invariants", action="store_true")
    args = ap.parse_args()

    import torch
    from boxtree_amd import HIPArrayContext, TreeBuilder
    from boxtree_amd.distributed.checksum import particle_order_checksum, tree_checksum
    from device_invariants import check_tree_on_device

    dev = torch.device("cuda", 0)
    actx = HIPArrayContext(0)
    tb = TreeBuilder(actx)
    wmax = max(args.worlds)
    n = args.n
    t0 = time.perf_counter()
    pts = [torch.empty(wmax * n, dtype=torch.float64, device=dev) for _ in range(3)]
    for g in range(wmax):
        rng = np.random.default_rng(15 + g)
        for ax in range(3):
            pts[ax][g * n:(g + 1) * n] = torch.from_numpy(rng.random(n)).to(dev)
    torch.cuda.synchronize()
    print(f"[c5_full] {wmax} chunks of {n} points generated and uploaded in "
          f"{time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)

    report = {}
    for world in sorted(args.worlds):
        m = world * n
        part = [p[:m] for p in pts]
        torch.cuda.synchronize()
        tree, _ = tb(actx, part, max_particles_in_box=args.mpb)     # warm-up (allocations)
        del tree
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        tree, ev = tb(actx, part, max_particles_in_box=args.mpb)
        ev.wait()
        torch.cuda.synchronize()
        build_ms = 1e3 * (time.perf_counter() - t1)
        nb = int(tree.nboxes)
        lsb = np.asarray(actx.to_numpy(tree.level_start_box_nrs), dtype=np.int64)
        cumul = tree.box_source_counts_cumul
        gids = torch.arange(nb, device=dev, dtype=torch.int64)
        entry = {
            "chunks": world, "points_per_chunk": n, "nsources": m,
            "max_particles_in_box": args.mpb,
            "nboxes": nb, "nlevels": int(tree.nlevels),
            "level_start_box_nrs": [int(v) for v in lsb],
            "counts_cumul_checksum": tree_checksum(torch, gids, cumul),
            "counts_cumul_sha256": hashlib.sha256(cumul.cpu().numpy().tobytes()).hexdigest(),
            # which particle sits where in tree order (sharded builds: the library's global ids)
            "user_source_ids_checksum": particle_order_checksum(torch, tree.user_source_ids),
            "root_extent": float(tree.root_extent),
            "bbox_min": [float(v) for v in tree.bounding_box[0]],
            "tree_build_ms_one_gpu": build_ms,
            "particles_per_s_tree_only": m / (build_ms * 1e-3),
        }
        if not args.no_invariants:
            t2 = time.perf_counter()
            inv = check_tree_on_device(torch, tree, part, args.mpb)
            entry["invariants"] = {**inv, "seconds": time.perf_counter() - t2,
                                   "what": "reference test_tree.py:88-220 restated on the device "
                                           "(tests/device_invariants.py)"}
        print(f"[c5_full] world {world}: {json.dumps(entry)}", file=sys.stderr, flush=True)
        report[str(world)] = entry
        del tree, cumul, gids
        torch.cuda.empty_cache()

    doc = {
        "what": "BASELINE configs[4] (3D uniform, 1.25e8 points per rank drawn with "
                "np.random.default_rng(15 + rank), max_particles_in_box 64): the tree ONE GPU builds "
                "from the chunks of W ranks, W = 1, 2, 4, 8 (8 = the 10^9-point tree).  Written by "
                "tools/c5_full.py on an MI355X after the tree passed the reference's assertions on "
                "the device; bench.py --gpus W --workload c5 compares its global numbering with it.",
        "user_source_ids_checksum": "sum_p (p + 1) * user_source_ids[p], wrapping int64 "
                                    "(checksum.particle_order_checksum): the ranks' sums over their "
                                    "slices of the tree order, with the library's global user ids, add "
                                    "up to it",
        "checksum": "sum_b box_source_counts_cumul[b] * ((b * 2654435761 mod 2^32) | 1), wrapping "
                    "int64 (boxtree_amd/distributed/checksum.py): linear in the counts, so the "
                    "ranks' checksums over their local trees add up to it",
        "worlds": report,
    }
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(doc, open(args.out, "w"), indent=1)
    if args.write:
        old = {}
        if os.path.exists(GOLDEN):
            old = json.load(open(GOLDEN)).get("worlds", {})
        if args.n == 125_000_000:
            old.update(report)
            doc["worlds"] = old
            json.dump(doc, open(GOLDEN, "w"), indent=1)
        else:
            print("[c5_full] --n is not the c5 chunk size: golden file not written", file=sys.stderr)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
