"""The sharded build as N real PROCESSES on the one GPU of a test box: every rank is a process of
its own (own HIP context, own bt_context) and drives the library's multi-GPU entries --
bt_mgpu_exchange, bt_tree_build, bt_mgpu_global_ids, bt_mgpu_number, bt_mgpu_let_build,
bt_traversal_build -- over the shared-memory communicator (bt_mgpu_comm_shm: RCCL refuses two ranks
on one device).  This is the code path bench.py times on N GPUs, not the torch twin the gloo tests
exercise.  What the ranks arrive at is compared with the ORACLE's single tree of the concatenated
chunks: global numbering, the tree checksum (linear in the per-box counts), the particle order by
the library's global user ids, and every interaction list through the row sums of
tests/sharded_sums.py (rows built by several ranks must agree)."""

import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def chunk(rank, n_per, dims, kind):
    rng = np.random.default_rng(15 + rank)
    if kind == "sphere":
        v = rng.standard_normal((dims, n_per))
        v /= np.sqrt((v * v).sum(axis=0))
        return [np.ascontiguousarray(v[i]) for i in range(dims)]
    return [rng.random(n_per) for _ in range(dims)]


def _rank_main(rank, world, name, n_per, dims, kind, mpb, slot_bytes, q):
    try:
        sys.path.insert(0, HERE)
        sys.path.insert(0, os.path.dirname(HERE))
        if os.environ.get("BOXTREE_EMU", "0") == "1":      # (tests/emu: this child is not a pytest process)
            sys.path.insert(0, os.path.join(HERE, "emu"))
            import emu_actx
            emu_actx.install_for_tests()
        import torch
        import sharded_sums as ss
        from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder
        from boxtree_amd.distributed import native as nat
        from boxtree_amd.distributed.checksum import particle_order_checksum, tree_checksum
        actx = HIPArrayContext(0)
        comm = nat.shm_comm(name, rank, world, slot_bytes=slot_bytes, timeout_s=120.0)
        mine = [torch.from_numpy(a).cuda() for a in chunk(rank, n_per, dims, kind)]
        p2, kw, xs = nat.exchange_particles(actx, comm, mine, mpb, own_buffer=True)
        tree, _ = TreeBuilder(actx)(actx, p2, max_particles_in_box=mpb, **kw)
        route = xs["route"]
        ids = route.global_user_source_ids(tree)
        # a per-particle array to the owners and back over the exchange's kept plan
        back = route.to_callers(route.to_owners(mine[0]))
        num = nat.number_sharded_tree(actx, comm, tree)
        let, info = nat.build_local_essential_tree(actx, comm, tree, num)
        trav, ev = FMMTraversalBuilder(actx)(actx, let, _target_boxes_mask=info["target_boxes_mask"],
                                             _active_level_ranges=info["active_level_ranges"])
        ev.wait()
        rows = ss.rank_rows(torch, trav, info["global_box_ids"], info["target_boxes_mask"])
        out = dict(
            rank=rank,
            checksum=tree_checksum(torch, num["box_ids"], tree.box_source_counts_cumul),
            ids_checksum=particle_order_checksum(torch, ids, num["source_offset"]),
            nboxes=int(num["nboxes"]), nlevels=int(num["nlevels"]),
            level_starts=[int(v) for v in num["global_level_start_box_nrs"]],
            nsources=int(tree.nsources), offset=int(num["source_offset"]),
            ntb=int(trav.target_boxes.shape[0]), halo=int(info["halo_boxes_received"]),
            round_trip=bool(torch.equal(back, mine[0])), rounds=int(xs["rounds"]),
            bytes_sent=int(xs["bytes_sent"]),
            rows={k: (g.cpu().numpy(), v.cpu().numpy()) for k, (g, v) in rows.items()})
        comm.close()
        q.put(out)
    except BaseException as e:      # noqa: BLE001
        import traceback
        q.put(dict(rank=rank, error=repr(e), trace=traceback.format_exc()[-2000:]))


def run_ranks(world, n_per, dims=3, kind="uniform", mpb=64, slot_bytes=64 << 20):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"/bt_mgpu_test_{os.getpid()}_{world}_{n_per}"
    procs = [ctx.Process(target=_rank_main, args=(r, world, name, n_per, dims, kind, mpb, slot_bytes, q))
             for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in range(world):
            res.append(q.get(timeout=600))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
        try:
            os.unlink("/dev/shm" + name)       # (the last rank removed it; a failed run may not have)
        except OSError:
            pass
    errors = [r for r in res if "error" in r]
    assert not errors, errors
    return sorted(res, key=lambda r: r["rank"])


@pytest.mark.gpu
@pytest.mark.parametrize("world,n_per,kind,slot_bytes", [
    (2, 1_000_000, "uniform", 64 << 20),
    (3, 400_000, "sphere", 1 << 20),          # 1-MiB slots: every all-to-all-v in many rounds
])
def test_process_ranks_equal_the_oracle_single_tree(oracle, world, n_per, kind, slot_bytes):
    import torch
    import fullsize_sums as fs
    import sharded_sums as ss
    res = run_ranks(world, n_per, kind=kind, slot_bytes=slot_bytes)
    pts = [np.concatenate([chunk(r, n_per, 3, kind)[ax] for r in range(world)]) for ax in range(3)]
    otree = oracle.build_tree(pts, max_particles_in_box=64)
    otrav = oracle.build_traversal(otree)
    nb = int(otree.nboxes)
    for r in res:
        assert (r["nboxes"], r["nlevels"]) == (nb, int(otree.nlevels))
        assert r["level_starts"] == [int(v) for v in otree.level_start_box_nrs]
        assert r["round_trip"] and r["halo"] > 0 and r["bytes_sent"] > 0
    assert sum(r["nsources"] for r in res) == world * n_per
    assert [r["offset"] for r in res] == [sum(q["nsources"] for q in res[:k]) for k in range(world)]
    gids = torch.arange(nb, dtype=torch.int64)
    assert fs.wrap(sum(r["checksum"] for r in res)) == fs.rows_sum(
        torch, gids, torch.from_numpy(otree.box_source_counts_cumul).to(torch.int64))
    assert fs.wrap(sum(r["ids_checksum"] for r in res)) == fs.array_sum(torch, otree.user_source_ids)
    assert sum(r["ntb"] for r in res) == len(otrav.target_boxes)
    merger = ss.RowMerger(torch, nb, int(otree.nlevels), "cpu")
    for r in res:
        merger.add({k: (torch.from_numpy(g), torch.from_numpy(v)) for k, (g, v) in r["rows"].items()})
    assert not merger.disagreements, sorted(set(merger.disagreements))
    got, want = merger.sums(), ss.single_tree_sums(torch, otree, otrav)
    assert got == want, fs.diff(got, want)
    if slot_bytes < (8 << 20):
        assert max(r["rounds"] for r in res) > 1


def test_shared_memory_collectives_cpu():
    """The communicator's collectives themselves, without a GPU: tests/cabi/shm_group_test (N forked
    processes, memcpy for the device copies, slots small enough for many rounds; sums, minima,
    ragged and empty messages, a size mismatch that must fail every rank instead of hanging)."""
    import subprocess
    exe = os.path.join(HERE, "cabi", "shm_group_test")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "cabi"), "shm_group_test"])
    for args in (["1", "8192"], ["2", "8192"], ["3", "4096"], ["8", "65536"]):
        p = subprocess.run([exe] + args, capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, (args, p.stdout, p.stderr)
