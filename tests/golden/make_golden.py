#!/usr/bin/env python
"""Generates the fixtures of tests/golden/*.npz.

The reference cannot be imported or compiled in the build container (DESIGN.md
section 2), so -- as SURVEY.md section 8(c) prescribes -- the fixtures are produced
by the CPU oracle (oracle/boxtree_oracle.c), each validated against every invariant
the reference's own tests assert before it is written.  They freeze the oracle:
tests/test_golden.py fails if the oracle's output for these inputs ever changes,
and the GPU tests compare the device output against the same files.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

TREE_FIELDS = [
    "level_start_box_nrs", "user_source_ids", "sorted_target_ids",
    "box_source_starts", "box_source_counts_nonchild", "box_source_counts_cumul",
    "box_target_starts", "box_target_counts_nonchild", "box_target_counts_cumul",
    "box_parent_ids", "box_child_ids", "box_centers", "box_levels", "box_flags",
    "box_source_bounding_box_min", "box_source_bounding_box_max",
    "box_target_bounding_box_min", "box_target_bounding_box_max",
]
TRAV_FIELDS = [
    "source_boxes", "target_boxes", "source_parent_boxes", "target_or_target_parent_boxes",
    "level_start_source_box_nrs", "level_start_target_box_nrs",
    "level_start_source_parent_box_nrs", "level_start_target_or_target_parent_box_nrs",
    "same_level_non_well_sep_boxes_starts", "same_level_non_well_sep_boxes_lists",
    "neighbor_source_boxes_starts", "neighbor_source_boxes_lists",
    "from_sep_siblings_starts", "from_sep_siblings_lists",
    "from_sep_bigger_starts", "from_sep_bigger_lists",
    "from_sep_close_smaller_starts", "from_sep_close_smaller_lists",
    "from_sep_close_bigger_starts", "from_sep_close_bigger_lists",
]

# name -> recipe (2D/3D, N in {4, 50, 1000}, mpb in {5, 30, 64}, separate
# targets, target radii with stick_out_factor 0 and 0.25)
CASES = {
    "2d_n4_mpb30": dict(dims=2, n=4, seed=1, kw=dict(max_particles_in_box=30)),
    "2d_n50_mpb5": dict(dims=2, n=50, seed=2, kw=dict(max_particles_in_box=5)),
    "2d_n1000_mpb30": dict(dims=2, n=1000, seed=3, kw=dict(max_particles_in_box=30)),
    "3d_n50_mpb5": dict(dims=3, n=50, seed=4, kw=dict(max_particles_in_box=5)),
    "3d_n1000_mpb5": dict(dims=3, n=1000, seed=5, kw=dict(max_particles_in_box=5)),
    "3d_n1000_mpb64": dict(dims=3, n=1000, seed=6, kw=dict(max_particles_in_box=64)),
    "3d_n4000_mpb30_f32": dict(dims=3, n=4000, seed=7, dtype="float32",
                               kw=dict(max_particles_in_box=30)),
    "2d_targets": dict(dims=2, n=600, nt=900, seed=8, kw=dict(max_particles_in_box=10)),
    "3d_targets_radii_so0": dict(dims=3, n=800, nt=500, seed=9, radii=True,
                                 kw=dict(max_particles_in_box=20, stick_out_factor=0.0)),
    "3d_targets_radii_so025": dict(dims=3, n=800, nt=500, seed=10, radii=True,
                                   kw=dict(max_particles_in_box=20, stick_out_factor=0.25)),
    "2d_level_restricted": dict(dims=2, n=1500, seed=11, clustered=True,
                                kw=dict(max_particles_in_box=10,
                                        kind="adaptive-level-restricted")),
    "3d_nway2": dict(dims=3, n=1500, seed=12, kw=dict(max_particles_in_box=10),
                     trav_kw=dict(well_sep_is_n_away=2)),
}


def make_inputs(case):
    rng = np.random.default_rng(case["seed"])
    dtype = np.dtype(case.get("dtype", "float64"))
    dims, n = case["dims"], case["n"]
    pts = [rng.standard_normal(n).astype(dtype) for _ in range(dims)]
    if case.get("clustered"):
        for p in pts:
            p[: n // 2] = 0.3 + 1e-3 * p[: n // 2]
    out = dict(particles=pts, targets=None, target_radii=None)
    if case.get("nt"):
        out["targets"] = [rng.standard_normal(case["nt"]).astype(dtype) for _ in range(dims)]
        if case.get("radii"):
            out["target_radii"] = (2.0 ** rng.uniform(-10, 0, case["nt"])).astype(dtype)
    return out


def build(oracle, case):
    inp = make_inputs(case)
    kw = dict(case["kw"])
    if inp["targets"] is not None:
        kw["targets"] = inp["targets"]
    if inp["target_radii"] is not None:
        kw["target_radii"] = inp["target_radii"]
    tree = oracle.build_tree(inp["particles"], **kw)
    trav = oracle.build_traversal(tree, **case.get("trav_kw", {}))
    return inp, tree, trav


def flatten(inp, tree, trav):
    d = {}
    for ax, p in enumerate(inp["particles"]):
        d[f"in_particles_{ax}"] = p
    if inp["targets"] is not None:
        for ax, p in enumerate(inp["targets"]):
            d[f"in_targets_{ax}"] = p
    if inp["target_radii"] is not None:
        d["in_target_radii"] = inp["target_radii"]
    d["tree_scalars"] = np.array([tree.nboxes, tree.nlevels, tree.aligned_nboxes], np.int64)
    d["tree_root_extent"] = np.asarray(tree.root_extent)
    d["tree_bbox"] = np.array([tree.bounding_box[0], tree.bounding_box[1]])
    for ax in range(tree.dimensions):
        d[f"tree_sources_{ax}"] = tree.sources[ax]
        d[f"tree_targets_{ax}"] = tree.targets[ax]
    for f in TREE_FIELDS:
        d["tree_" + f] = getattr(tree, f)
    for f in TRAV_FIELDS:
        v = getattr(trav, f)
        if v is not None:
            d["trav_" + f] = v
    for lev, bl in enumerate(trav.from_sep_smaller_by_level):
        d[f"trav_l3_{lev}_starts"] = bl.starts
        d[f"trav_l3_{lev}_lists"] = bl.lists
        d[f"trav_l3_{lev}_nonempty_indices"] = bl.nonempty_indices
        d[f"trav_l3_{lev}_compressed_indices"] = bl.compressed_indices
        d[f"trav_l3_{lev}_target_boxes"] = trav.target_boxes_sep_smaller_by_source_level[lev]
    return d


def main():
    from invariants import check_traversal, check_tree, constant_one_potentials
    from oracle import oracle
    oracle.build_lib()
    for name, case in CASES.items():
        inp, tree, trav = build(oracle, case)
        kw = {k: v for k, v in case["kw"].items()
              if k in ("max_particles_in_box",)}
        if case["kw"].get("kind", "adaptive") == "adaptive":
            check_tree(tree, inp["particles"], targets=inp["targets"],
                       target_radii=inp["target_radii"], **kw)
        if not case.get("trav_kw"):       # test_tree_connectivity only covers n-away = 1
            check_traversal(tree, trav)
        pot = constant_one_potentials(tree, trav)
        assert np.all(pot == tree.nsources), name
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **flatten(inp, tree, trav))
        print(f"{name}: {tree.nboxes} boxes, {tree.nlevels} levels")


if __name__ == "__main__":
    main()
