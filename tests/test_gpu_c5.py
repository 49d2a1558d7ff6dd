"""BASELINE configs[4] -- 3D uniform, 10^9 points, 8 chunks drawn with default_rng(15 + g) --
at FULL size on the one GPU a test box has.

* the tree of all 8 chunks (tree only: its List 2 exceeds the reference's int32 CSR) is
  built by one GPU, passes the reference's tree assertions restated on the device
  (tests/device_invariants.py <- test/test_tree.py:88-220) and reproduces the committed
  counts and checksum (tests/golden/c5_global_counts.json, written by tools/c5_full.py);
* a rank's share: two ranks (threads of this process, the library's own multi-rank entries)
  hold one chunk each at full chunk size, exchange, build, number globally, assemble their
  local essential trees and build their lists; the global numbering they arrive at is the
  committed single-GPU tree of chunks 0 and 1 -- the comparison bench.py --gpus N makes
  (config.c5_check).
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


def test_c5_share_of_two_ranks_numbers_like_the_single_gpu_tree():
    import torch
    from boxtree_amd import HIPArrayContext
    from boxtree_amd.distributed import native as nat
    from boxtree_amd.distributed.checksum import tree_checksum, wrap_int64
    world = 2
    free, _total = torch.cuda.mem_get_info(0)
    if free < world * N_CHUNK * 400:
        pytest.skip("needs ~100 GB of free device memory")
    pts = upload_chunks(torch, world)
    group = nat.LocalGroup(world)
    res, errors = [None] * world, []

    def run(rank):
        try:
            actx = HIPArrayContext(0)
            comm = group.comm(rank)
            mine = [p[rank * N_CHUNK:(rank + 1) * N_CHUNK] for p in pts]
            out = nat.sharded_tree_and_lists(actx, comm, mine, 64)
            num, tree = out["numbering"], out["tree"]
            res[rank] = dict(
                checksum=tree_checksum(torch, num["box_ids"], tree.box_source_counts_cumul),
                nboxes=num["nboxes"], nlevels=num["nlevels"],
                level_starts=[int(v) for v in num["global_level_start_box_nrs"]],
                nsources=int(tree.nsources), offset=num["source_offset"],
                lists=int(out["traversal"].from_sep_siblings_lists.shape[0]),
                halo=out["let_info"]["halo_boxes_received"])
            comm.close()
            actx.lib.bt_trim(actx.handle)
        except BaseException as e:      # noqa: BLE001
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert all(not t.is_alive() for t in threads), "a rank hangs"
    group.close()
    assert not errors, errors
    gold = GOLDEN[str(world)]
    assert wrap_int64(sum(r["checksum"] for r in res)) == gold["counts_cumul_checksum"]
    for r in res:
        assert r["nboxes"] == gold["nboxes"] and r["nlevels"] == gold["nlevels"]
        assert r["level_starts"] == gold["level_start_box_nrs"]
        assert r["lists"] > 0 and r["halo"] > 0
    assert sum(r["nsources"] for r in res) == world * N_CHUNK
    assert res[0]["offset"] == 0 and res[1]["offset"] == res[0]["nsources"]
