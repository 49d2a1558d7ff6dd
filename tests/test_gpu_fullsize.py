"""BASELINE.json's configurations at FULL size against the CPU oracle.

The oracle ran in the build container (tests/golden/make_fullsize_oracle_sums.py: 10^7 ... 2.5 * 10^8
points, minutes each) and left one checksum per Tree / FMMTraversalInfo array in
tests/golden/fullsize_oracle_sums.json (tests/fullsize_sums.py: position-weighted wrapping sums
over the arrays' bit patterns).  Here the HIP path builds the same configurations from the same
host-drawn inputs and must arrive at the same sums, array by array -- the comparison
tests/compare.py makes at oracle-sized samples, at the sizes bench.py times.  Nothing here was
written by the product.

  c1 c2 c3 c3c c4   tree + every list                              c5w1  one rank's chunk (1.25e8)
  c5w2  the tree of two chunks (2.5e8; lists exceed int32 CSR)       c5r8  eight ranks x 15 625 000:
                                                                          every rank's lists, summed
"""

import json
import os
import threading

import numpy as np
import pytest

import fullsize_sums as fs

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "fullsize_oracle_sums.json")) as _f:
    GOLDEN = json.load(_f)["configs"]


def golden_inputs(cfg):
    """The recipe of tests/golden/make_fullsize_oracle_sums.py::inputs, on the host."""
    if "workload" in cfg:
        from bench import make_workload_numpy
        return make_workload_numpy(cfg["workload"], cfg["n"], 15)
    parts = [[], [], []]
    for g in range(cfg["chunks"]):
        rng = np.random.default_rng(15 + g)
        for ax in range(3):
            parts[ax].append(rng.random(cfg["n_chunk"]))
    return dict(particles=[np.concatenate(p) for p in parts], targets=None, kw={})


EXTRAS = ("counts_cumul_checksum", "user_source_ids_checksum", "level_start_box_nrs.values")


def assert_sums(got, want, what):
    """every array's sum, and every scalar, equals the oracle's (EXTRAS: entries of the golden file
    that the product's own checksum functions are compared with separately)"""
    bad = fs.diff(got, {k: v for k, v in want.items() if k not in EXTRAS})
    assert not bad, f"{what}: {len(bad)} arrays differ from the oracle's: {bad[:12]}"


def release(torch, actx):
    torch.cuda.empty_cache()
    actx.lib.bt_trim(actx.handle)


@pytest.mark.parametrize("name", ["c1", "c2", "c3", "c3c", "c4", "c5w1", "c5w2"])
def test_full_size_configuration_equals_the_oracle(name):
    import torch
    from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder
    if name not in GOLDEN:
        pytest.skip(f"no oracle sums for {name} in tests/golden/fullsize_oracle_sums.json")
    gold = GOLDEN[name]
    cfg = gold["config"]
    n_total = cfg.get("n", cfg.get("chunks", 1) * cfg.get("n_chunk", 0))
    free, _total = torch.cuda.mem_get_info(0)
    if free < n_total * 700:
        pytest.skip(f"needs ~{n_total * 700 >> 30} GB of free device memory")
    actx = HIPArrayContext(0)
    w = golden_inputs(cfg)
    up = lambda arrs: None if arrs is None else [torch.from_numpy(a).cuda() for a in arrs]  # noqa: E731
    kw = dict(w["kw"])
    if "target_radii" in kw:
        kw["target_radii"] = torch.from_numpy(kw["target_radii"]).cuda()
    pts, tgts = up(w["particles"]), up(w["targets"])
    del w
    tree, ev = TreeBuilder(actx)(actx, pts, targets=tgts, max_particles_in_box=cfg["mpb"], **kw)
    ev.wait()
    got = fs.tree_sums(torch, tree)
    assert_sums(got, gold["tree"], f"{name} tree")
    # the two sums sharded builds and bench.py's c5_check use, by the product's own functions
    from boxtree_amd.distributed.checksum import particle_order_checksum, tree_checksum
    ids = torch.arange(int(tree.nboxes), device=tree.box_source_counts_cumul.device)
    assert tree_checksum(torch, ids, tree.box_source_counts_cumul) == gold["tree"]["counts_cumul_checksum"]
    assert particle_order_checksum(torch, tree.user_source_ids) == gold["tree"]["user_source_ids_checksum"]
    if "traversal" in gold:
        trav, ev = FMMTraversalBuilder(actx)(actx, tree)
        ev.wait()
        assert_sums(fs.traversal_sums(torch, trav), gold["traversal"], f"{name} traversal")
        del trav
    del tree, pts, tgts, kw
    release(torch, actx)


def sharded_lists_against_single_tree(world, n_chunk, mpb, gold):
    """`world` ranks (threads over the library's local communicator) with the c5 recipe's chunks of
    n_chunk points: exchange, build, number globally, local essential trees, lists; the ranks' rows,
    mapped to global box numbers, must add up to *gold* (the single tree's sums, tests/sharded_sums.py;
    checksums of the tree and of the particle order as in distributed/checksum.py)."""
    import torch
    import sharded_sums as ss
    from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder
    from boxtree_amd.distributed import native as nat
    from boxtree_amd.distributed.checksum import particle_order_checksum, tree_checksum, wrap_int64
    nglobal = gold["nboxes"]
    dev = torch.device("cuda", 0)
    group = nat.LocalGroup(world)
    res, errors = [None] * world, []
    merger = ss.RowMerger(torch, nglobal, gold["nlevels"], dev)
    merge = threading.Lock()

    def run(rank):
        try:
            actx = HIPArrayContext(0)
            comm = group.comm(rank)
            rng = np.random.default_rng(15 + rank)
            mine = [torch.from_numpy(rng.random(n_chunk)).cuda() for _ in range(3)]
            p2, kw, xs = nat.exchange_particles(actx, comm, mine, mpb, own_buffer=True)
            tree, _ = TreeBuilder(actx)(actx, p2, max_particles_in_box=mpb, **kw)
            ids = xs["route"].global_user_source_ids(tree)
            num = nat.number_sharded_tree(actx, comm, tree)
            let, info = nat.build_local_essential_tree(actx, comm, tree, num)
            trav, ev = FMMTraversalBuilder(actx)(actx, let, _target_boxes_mask=info["target_boxes_mask"],
                                                 _active_level_ranges=info["active_level_ranges"])
            ev.wait()
            rows = ss.rank_rows(torch, trav, info["global_box_ids"], info["target_boxes_mask"])
            with merge:
                merger.add(rows)
            res[rank] = dict(
                checksum=tree_checksum(torch, num["box_ids"], tree.box_source_counts_cumul),
                ids_checksum=particle_order_checksum(torch, ids, num["source_offset"]),
                nboxes=num["nboxes"], nlevels=num["nlevels"],
                level_starts=[int(v) for v in num["global_level_start_box_nrs"]],
                ntb=int(trav.target_boxes.shape[0]), nlev3=len(trav.from_sep_smaller_by_level))
            comm.close()
        except BaseException as e:      # noqa: BLE001
            import traceback
            errors.append((rank, repr(e), traceback.format_exc()[-1500:]))

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=900)
    assert all(not t.is_alive() for t in threads), "a rank hangs"
    group.close()
    assert not errors, errors
    assert not merger.disagreements, f"ranks disagree on shared rows of {sorted(set(merger.disagreements))}"
    for r in res:
        assert (r["nboxes"], r["nlevels"]) == (gold["nboxes"], gold["nlevels"])
        assert r["level_starts"] == gold["level_start_box_nrs"]
        assert r["nlev3"] <= gold["nlevels"]
    assert wrap_int64(sum(r["checksum"] for r in res)) == gold["counts_cumul_checksum"]
    assert wrap_int64(sum(r["ids_checksum"] for r in res)) == gold["user_source_ids_checksum"]
    assert sum(r["ntb"] for r in res) == gold["ntarget_boxes"]       # every target box: exactly one rank
    got = merger.sums()
    want = {"colleagues": gold["colleagues"], "list1": gold["list1"], "list2": gold["list2"],
            "list4": gold["list4"], **{f"list3[{lev}]": v for lev, v in enumerate(gold["list3"])}}
    assert got == want, fs.diff(got, want)


def test_c5_eight_ranks_lists_sum_to_the_oracle_single_tree():
    """BASELINE configs[4]'s split over N = 8 at 8 x 15 625 000 points (the largest size whose
    single-tree lists fit the reference's int32 CSR): eight ranks (threads over the library's local
    communicator) exchange, build, number globally, assemble their local essential trees and build
    their lists; per list, the rows every rank built -- box numbers mapped to global ones -- carry
    the values of the ORACLE's single tree: sum_rows w(box) * sum_k (k + 1) * (entry_k + 1).  Rows
    that several ranks build (shared top boxes) must agree; a row nobody builds counts as empty."""
    if "c5r8" not in GOLDEN:
        pytest.skip("no oracle sums for c5r8")
    cfg = GOLDEN["c5r8"]["config"]
    sharded_lists_against_single_tree(cfg["chunks"], cfg["n_chunk"], cfg["mpb"], GOLDEN["c5r8"]["sharded"])


@pytest.mark.parametrize("world,n_chunk", [(8, 60_000), (3, 150_000)])
def test_sharded_lists_sum_to_the_oracle_single_tree_small(oracle, world, n_chunk):
    """The same comparison at a size the oracle builds on the spot (and the CPU emulation of the
    kernels gets through: tests/emu): the sums are made from the oracle's single tree here."""
    import torch
    sys_path_golden = os.path.join(HERE, "golden")
    import sys
    if sys_path_golden not in sys.path:
        sys.path.insert(0, sys_path_golden)
    import make_fullsize_oracle_sums as mk
    parts = [[], [], []]
    for g in range(world):
        rng = np.random.default_rng(15 + g)
        for ax in range(3):
            parts[ax].append(rng.random(n_chunk))
    pts = [np.concatenate(p) for p in parts]
    otree = oracle.build_tree(pts, max_particles_in_box=64)
    gold = mk.sharded_sums(torch, otree, oracle.build_traversal(otree))
    sharded_lists_against_single_tree(world, n_chunk, 64, gold)
