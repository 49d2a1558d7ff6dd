"""CPU-side checks of the full-size oracle checksums (tests/golden/fullsize_oracle_sums.json):
the file is what the committed maker script produces, the checksums see what tests/compare.py sees,
and the c5 golden file the PRODUCT wrote on a GPU (tests/golden/c5_global_counts.json) agrees with
what the ORACLE builds at the same size."""

import json
import os
import sys

import numpy as np
import pytest

import fullsize_sums as fs

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "fullsize_oracle_sums.json")) as _f:
    GOLDEN = json.load(_f)["configs"]


def test_c5_golden_written_by_the_product_is_what_the_oracle_builds():
    """tests/golden/c5_global_counts.json (worlds 1 and 2) was written by the product on a GPU
    (tools/c5_full.py, rounds 4 and 5); the oracle, run later in the build container, arrives at the same box counts,
    level starts, counts checksum and particle-order checksum."""
    with open(os.path.join(HERE, "golden", "c5_global_counts.json")) as f:
        prod = json.load(f)["worlds"]
    for world, name in (("1", "c5w1"), ("2", "c5w2")):
        if name not in GOLDEN:
            continue
        o, p = GOLDEN[name]["tree"], prod[world]
        assert (o["nboxes"], o["nlevels"]) == (p["nboxes"], p["nlevels"])
        assert o["level_start_box_nrs.values"] == p["level_start_box_nrs"]
        assert o["counts_cumul_checksum"] == p["counts_cumul_checksum"]
        assert o["user_source_ids_checksum"] == p["user_source_ids_checksum"]
        assert float.fromhex(o["root_extent_hex"]) == p["root_extent"]



def test_golden_file_is_reproducible_c1():
    """Re-running the maker's recipe for c1 (2D uniform 10^5, the reference's CPU-runnable
    configuration) gives the committed sums: the file is the script's output, unedited."""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_fullsize_oracle_sums as mk
    entry = mk.run_one("c1")
    assert entry["tree"] == GOLDEN["c1"]["tree"]
    assert entry["traversal"] == GOLDEN["c1"]["traversal"]


def test_sums_are_position_and_bit_sensitive():
    import torch
    a = np.arange(1, 11, dtype=np.int32)
    b = a.copy()
    b[[2, 7]] = b[[7, 2]]
    assert fs.array_sum(torch, a) != fs.array_sum(torch, b)                 # a swap
    assert fs.array_sum(torch, a) == fs.array_sum(torch, torch.from_numpy(a))
    x = np.array([0.1, 0.2, 0.30000000000000004])
    y = np.array([0.1, 0.2, 0.3])
    assert fs.array_sum(torch, x) != fs.array_sum(torch, y)                 # one ulp
    # rows: order within a row counts, rows may be summed in pieces
    starts = np.array([0, 3, 3, 5], np.int32)
    lists = np.array([4, 9, 2, 7, 1], np.int32)
    swapped = np.array([9, 4, 2, 7, 1], np.int32)
    assert fs.csr_rows_sum(torch, starts, lists) != fs.csr_rows_sum(torch, starts, swapped)
    vals = fs.csr_row_values(torch, starts, lists)
    assert vals.tolist() == [1 * 5 + 2 * 10 + 3 * 3, 0, 1 * 8 + 2 * 2]
    g = torch.arange(3)
    whole = fs.rows_sum(torch, g, vals)
    assert fs.wrap(fs.rows_sum(torch, g[:1], vals[:1]) + fs.rows_sum(torch, g[1:], vals[1:])) == whole
    # entries mapped to global numbers
    gid = np.arange(10, dtype=np.int64) * 3
    assert fs.csr_row_values(torch, starts, lists, entry_gid=gid).tolist() == [
        1 * 13 + 2 * 28 + 3 * 7, 0, 1 * 22 + 2 * 4]


def test_tree_sums_equal_iff_compare_passes(oracle):
    """On a small oracle tree: identical builds give identical sums; one changed element of any
    array changes that array's sum."""
    import torch
    rng = np.random.default_rng(3)
    pts = [rng.random(3000) for _ in range(3)]
    t1 = oracle.build_tree(pts, max_particles_in_box=20)
    t2 = oracle.build_tree(pts, max_particles_in_box=20)
    s1, s2 = fs.tree_sums(torch, t1), fs.tree_sums(torch, t2)
    assert s1 == s2
    t2.box_centers[1, 5] = np.nextafter(t2.box_centers[1, 5], 2.0)
    t2.user_source_ids[[0, 1]] = t2.user_source_ids[[1, 0]]
    assert fs.diff(fs.tree_sums(torch, t2), s1) == ["box_centers", "user_source_ids"]
    v1 = fs.traversal_sums(torch, oracle.build_traversal(t1))
    tr = oracle.build_traversal(t1)
    tr.from_sep_siblings_lists[10] += 1
    assert fs.diff(fs.traversal_sums(torch, tr), v1) == ["from_sep_siblings_lists"]
