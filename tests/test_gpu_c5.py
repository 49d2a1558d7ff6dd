"""BASELINE configs[4] -- 3D uniform, 10^9 points, 8 chunks drawn with default_rng(15 + g) --
at FULL size on the one GPU a test box has.

* the tree of all 8 chunks (tree only: its List 2 exceeds the reference's int32 CSR) is
  built by one GPU, passes the reference's tree assertions restated on the device
  (tests/device_invariants.py <- test/test_tree.py:88-220) and reproduces the committed
  counts and checksum (tests/golden/c5_global_counts.json, written by tools/c5_full.py);
* a rank's share: 2, 4 and 8 ranks (threads of this process, the library's own multi-rank
  entries) hold one chunk each at full chunk size, exchange, build, number globally, assemble
  their local essential trees and build their lists; the global numbering they arrive at and
  the order of their particles (the library's global user ids) are those of the committed
  single-GPU tree of the same chunks -- the comparison bench.py --gpus N makes
  (config.c5_check);
* the lists themselves at N = 8: the c5 recipe at 2 * 10^6 points per rank, every list of every
  rank against the single-GPU traversal.
"""

import json
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "c5_global_counts.json")))["worlds"]
N_CHUNK = 125_000_000


def upload_chunks(torch, world):
    dev = torch.device("cuda", 0)
    pts = [torch.empty(world * N_CHUNK, dtype=torch.float64, device=dev) for _ in range(3)]
    for g in range(world):
        rng = np.random.default_rng(15 + g)
        for ax in range(3):
            pts[ax][g * N_CHUNK:(g + 1) * N_CHUNK] = torch.from_numpy(rng.random(N_CHUNK)).to(dev)
    return pts


def assert_golden(torch, actx, tree, gold):
    from boxtree_amd.distributed.checksum import tree_checksum
    nb = int(tree.nboxes)
    assert nb == gold["nboxes"] and int(tree.nlevels) == gold["nlevels"]
    assert [int(v) for v in actx.to_numpy(tree.level_start_box_nrs)] == gold["level_start_box_nrs"]
    ids = torch.arange(nb, device=tree.box_source_counts_cumul.device)
    assert tree_checksum(torch, ids, tree.box_source_counts_cumul) == gold["counts_cumul_checksum"]


@pytest.mark.parametrize("world", [1, 8])
def test_c5_tree_of_all_chunks_on_one_gpu(world):
    import torch
    from boxtree_amd import HIPArrayContext, TreeBuilder
    from device_invariants import check_tree_on_device
    free, _total = torch.cuda.mem_get_info(0)
    if free < world * N_CHUNK * 200:
        pytest.skip(f"needs ~{world * N_CHUNK * 200 >> 30} GB of free device memory")
    actx = HIPArrayContext(0)
    pts = upload_chunks(torch, world)
    tree, ev = TreeBuilder(actx)(actx, pts, max_particles_in_box=64)
    ev.wait()
    assert_golden(torch, actx, tree, GOLDEN[str(world)])
    rep = check_tree_on_device(torch, tree, pts, 64)
    assert rep["nleaves"] == GOLDEN[str(world)]["invariants"]["nleaves"]
    del tree, pts
    torch.cuda.empty_cache()
    actx.lib.bt_trim(actx.handle)


def test_c5_recipe_eight_ranks_lists_are_the_single_gpu_traversal():
    """The N = 8 dress rehearsal of BASELINE configs[4] on one GPU: eight ranks (threads over the
    library's local communicator), chunk g = 3 x default_rng(15 + g).random(2 * 10^6) -- the c5
    recipe at a size whose single-GPU TRAVERSAL fits the reference's int32 lists --, the
    library's default ownership level (5), max_particles_in_box 64: every rank's local essential
    tree is the global tree restricted to its boxes, and every list of every rank (colleagues,
    Lists 1-4 per level) equals the rows of the single-GPU traversal; every target box's lists are
    built by exactly one rank."""
    from test_gpu_parity import check_multi_rank_let
    check_multi_rank_let(3, 8, "uniform", 1, n_per=2_000_000, mpb=64, top_level=5, seed=15,
                         expect_partial=True, native=True)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_c5_share_of_n_ranks_numbers_like_the_single_gpu_tree(world):
    """`world` ranks (threads), one FULL chunk of 1.25 * 10^8 points each: exchange, per-rank
    build, global numbering, particle identity, local essential tree and lists.  The global
    numbering the ranks arrive at and the order of their particles -- named by the library's global
    user ids -- are those of the tree ONE GPU builds from all `world` chunks (world 8: the
    10^9-point tree), by the committed checksums.  The ranks' builds take turns on the one GPU
    (a lock around the local build and the list build, pools trimmed in between): the collectives
    are what needs every rank present."""
    import torch
    from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder
    from boxtree_amd.distributed import native as nat
    from boxtree_amd.distributed.checksum import particle_order_checksum, tree_checksum, wrap_int64
    free, _total = torch.cuda.mem_get_info(0)
    # inputs 24 B + received 24 B + tree 40 B + lists ~60 B per particle and rank, one build in flight
    need = world * N_CHUNK * 160 + N_CHUNK * 400
    if free < need:
        pytest.skip(f"needs ~{need >> 30} GB of free device memory")
    pts = upload_chunks(torch, world)
    group = nat.LocalGroup(world)
    res, errors = [None] * world, []
    turn = threading.Lock()

    def run(rank):
        try:
            actx = HIPArrayContext(0)
            comm = group.comm(rank)
            mine = [p[rank * N_CHUNK:(rank + 1) * N_CHUNK] for p in pts]
            p2, kw, xs = nat.exchange_particles(actx, comm, mine, 64, own_buffer=True)
            actx.synchronize()
            actx.lib.bt_release_cached(actx.handle)     # (the send buffer and the cell indices)
            with turn:
                tree, _ = TreeBuilder(actx)(actx, p2, max_particles_in_box=64, **kw)
                actx.synchronize()
                actx.lib.bt_release_cached(actx.handle)
            ids = xs["route"].global_user_source_ids(tree)
            num = nat.number_sharded_tree(actx, comm, tree)
            let, info = nat.build_local_essential_tree(actx, comm, tree, num)
            with turn:
                trav, _ = FMMTraversalBuilder(actx)(actx, let, _target_boxes_mask=info["target_boxes_mask"],
                                                    _active_level_ranges=info["active_level_ranges"])
                actx.synchronize()
                nlist2 = int(trav.from_sep_siblings_lists.shape[0])
                ntb = int(trav.target_boxes.shape[0])
                del trav
                actx.lib.bt_release_cached(actx.handle)
            res[rank] = dict(
                checksum=tree_checksum(torch, num["box_ids"], tree.box_source_counts_cumul),
                ids_checksum=particle_order_checksum(torch, ids, num["source_offset"]),
                nboxes=num["nboxes"], nlevels=num["nlevels"],
                level_starts=[int(v) for v in num["global_level_start_box_nrs"]],
                nsources=int(tree.nsources), offset=num["source_offset"], lists=nlist2, ntb=ntb,
                halo=info["halo_boxes_received"], chunk_offset=xs["route"].chunk_offset["sources"])
            comm.close()
            del tree, let, info, ids, p2, xs
            torch.cuda.empty_cache()
            actx.lib.bt_trim(actx.handle)
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
    gold = GOLDEN[str(world)]
    assert wrap_int64(sum(r["checksum"] for r in res)) == gold["counts_cumul_checksum"]
    assert wrap_int64(sum(r["ids_checksum"] for r in res)) == gold["user_source_ids_checksum"]
    for k, r in enumerate(res):
        assert r["nboxes"] == gold["nboxes"] and r["nlevels"] == gold["nlevels"]
        assert r["level_starts"] == gold["level_start_box_nrs"]
        assert r["lists"] > 0 and r["halo"] > 0 and r["ntb"] > 0
        assert r["chunk_offset"] == k * N_CHUNK
    assert sum(r["nsources"] for r in res) == world * N_CHUNK
    assert [r["offset"] for r in res] == [sum(q["nsources"] for q in res[:k]) for k in range(world)]
    del pts
    torch.cuda.empty_cache()
