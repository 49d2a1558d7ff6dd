"""The C ABI used from plain C: tests/cabi/cabi_tree.c (no Python, no torch in the
program) builds a tree through include/boxtree_hip.h; its result must be the tree the
Python layer builds from the same points."""

import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "cabi", "cabi_tree")


def splitmix64_uniform(seed, n):
    """The generator of cabi_tree.c, vectorised: the k-th draw uses state seed + k*G."""
    with np.errstate(over="ignore"):
        g = np.uint64(0x9e3779b97f4a7c15)
        z = np.uint64(seed) + g * np.arange(1, n + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


@pytest.mark.parametrize("dims,n,mpb,seed", [(3, 200000, 30, 7), (2, 50000, 5, 11),
                                             (3, 1000000, 64, 3), (1, 3000, 10, 5)])
def test_plain_c_program_builds_the_same_tree(dims, n, mpb, seed):
    if not os.path.exists(EXE):
        pytest.fail("tests/cabi/cabi_tree missing: __graft_entry__.build() compiles it")
    out = subprocess.run([EXE, str(dims), str(n), str(mpb), str(seed)], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1000:]
    tok = out.stdout.split()
    got = {tok[i]: tok[i + 1] for i in range(0, len(tok), 2)}

    from boxtree_amd import HIPArrayContext, TreeBuilder
    actx = HIPArrayContext(0)
    # axis ax draws n consecutive values from the one stream
    pts = [splitmix64_uniform(seed, n * dims)[ax * n:(ax + 1) * n] for ax in range(dims)]
    tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(p) for p in pts],
                                max_particles_in_box=mpb)
    h = actx.to_numpy(tree)
    assert int(got["nboxes"]) == h.nboxes and int(got["nlevels"]) == h.nlevels
    assert int(got["aligned"]) == h.aligned_nboxes
    assert float(got["root_extent"]) == float(h.root_extent)
    nb = h.nboxes
    cumul = int((h.box_source_counts_cumul.astype(np.uint64)
                 * np.arange(1, nb + 1, dtype=np.uint64)).sum())
    ids = int((h.user_source_ids.astype(np.uint64)
               * (np.arange(n, dtype=np.uint64) % np.uint64(1000003) + np.uint64(1))).sum())
    assert int(got["cumul"]) == cumul
    assert int(got["ids"]) == ids
    assert int(got["levels"]) == int(h.box_levels.astype(np.uint64).sum())


@pytest.mark.parametrize("nranks,dims,n,mpb,seed", [(1, 3, 100000, 30, 21), (3, 3, 80000, 30, 5),
                                                   (4, 2, 60000, 10, 9), (8, 3, 40000, 64, 2)])
def test_plain_c_program_runs_the_sharded_build(nranks, dims, n, mpb, seed):
    """tests/cabi/cabi_mgpu.c: steps 1-6 of the sharded build (exchange, per-rank build,
    global numbering, local essential tree) from plain C, the ranks being pthreads over the
    library's local communicator.  The global figures every rank reports must be those of the
    tree one GPU builds from all chunks; the ranks' deep boxes carry every global number
    below the shared top levels exactly once."""
    exe = os.path.join(HERE, "cabi", "cabi_mgpu")
    if not os.path.exists(exe):
        pytest.fail("tests/cabi/cabi_mgpu missing: __graft_entry__.build() compiles it")
    out = subprocess.run([exe, str(nranks), str(dims), str(n), str(mpb), str(seed)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-1500:]
    rows = []
    for line in out.stdout.splitlines():
        tok = line.split()
        if tok and tok[0] == "rank":
            rows.append({tok[i]: int(tok[i + 1]) for i in range(0, len(tok), 2)})
    assert [r["rank"] for r in rows] == list(range(nranks))

    from boxtree_amd import HIPArrayContext, TreeBuilder
    actx = HIPArrayContext(0)
    chunks = [[splitmix64_uniform(seed + r, n * dims)[ax * n:(ax + 1) * n] for ax in range(dims)]
              for r in range(nranks)]
    pts = [np.concatenate([c[ax] for c in chunks]) for ax in range(dims)]
    tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(p) for p in pts], max_particles_in_box=mpb)
    h = actx.to_numpy(tree)
    top_level = 5 if dims == 3 else 7
    assert sum(r["owned"] for r in rows) == nranks * n
    offsets = np.cumsum([0] + [r["owned"] for r in rows[:-1]])
    for r, off in zip(rows, offsets):
        assert r["nboxes_global"] == h.nboxes and r["nlevels_global"] == h.nlevels
        assert r["nsources_global"] == nranks * n and r["source_offset"] == int(off)
        assert r["let_nboxes"] <= h.nboxes
    if nranks == 1:
        assert rows[0]["let_nboxes"] == h.nboxes and rows[0]["halo_in"] == 0
    deep = np.nonzero(h.box_levels > top_level)[0].astype(np.uint64)
    assert sum(r["deep_ids"] for r in rows) == int(deep.sum())
    # particle identity from C: sum over tree positions p of (p + 1) * (global user id of the source
    # at p), over all ranks, is the single-GPU tree's (mod 2^64) -- every rank's ids, in its slice
    want = int((np.arange(1, h.nsources + 1, dtype=np.uint64) * h.user_source_ids.astype(np.uint64)).sum())
    assert sum(r["user_id_digest"] for r in rows) % 2**64 == want
    assert [r["chunk_offset"] for r in rows] == [k * n for k in range(nranks)]


@pytest.mark.parametrize("nranks,dims,n,nt,mpb,seed", [(2, 3, 60000, 12000, 30, 4), (5, 2, 30000, 8000, 10, 8),
                                                      (3, 3, 40000, 9000, 16, 12)])
def test_plain_c_program_runs_the_sharded_build_with_target_extents(nranks, dims, n, nt, mpb, seed):
    """The same program with separate targets that have extents (bt_mgpu_params.target_radii,
    the shard's record length and radii column, bt_tree_params.top_box_arrive / _stay, the three
    extra arrays of the local tree view and of the LET export): global box, level and particle
    counts, particle offsets and the global numbers of all owned deep boxes are those of the tree
    one GPU builds from all chunks with target_radii."""
    exe = os.path.join(HERE, "cabi", "cabi_mgpu")
    if not os.path.exists(exe):
        pytest.fail("tests/cabi/cabi_mgpu missing: __graft_entry__.build() compiles it")
    out = subprocess.run([exe, str(nranks), str(dims), str(n), str(mpb), str(seed), str(nt)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-1500:]
    rows = []
    for line in out.stdout.splitlines():
        tok = line.split()
        if tok and tok[0] == "rank":
            rows.append({tok[i]: int(tok[i + 1]) for i in range(0, len(tok), 2)})
    assert [r["rank"] for r in rows] == list(range(nranks))

    from boxtree_amd import HIPArrayContext, TreeBuilder
    actx = HIPArrayContext(0)
    src = [[splitmix64_uniform(seed + r, n * dims)[ax * n:(ax + 1) * n] for ax in range(dims)]
           for r in range(nranks)]
    tgt, rad = [], []
    for r in range(nranks):
        draws = splitmix64_uniform(seed + 1000 + r, nt * (dims + 1))
        tgt.append([draws[ax * nt:(ax + 1) * nt] for ax in range(dims)])
        u = draws[dims * nt:]
        k = np.floor(12.0 * u)
        rad.append((1.0 / 2.0 ** (4 + k)) * (1.0 - 0.5 * (12.0 * u - k)))
    cat = lambda chunks: [np.concatenate([c[ax] for c in chunks]) for ax in range(dims)]  # noqa: E731
    tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(p) for p in cat(src)],
                                targets=[actx.from_numpy(p) for p in cat(tgt)],
                                target_radii=actx.from_numpy(np.concatenate(rad)), stick_out_factor=0.25,
                                max_particles_in_box=mpb)
    h = actx.to_numpy(tree)
    top_level = 5 if dims == 3 else 7
    # the case must have targets that stay in boxes of the shared top levels
    assert np.any((h.box_levels <= top_level) & (h.box_target_counts_nonchild > 0)
                  & (h.box_target_counts_nonchild < h.box_target_counts_cumul))
    assert sum(r["owned"] for r in rows) == nranks * n
    assert sum(r["targets_owned"] for r in rows) == nranks * nt
    soff = np.cumsum([0] + [r["owned"] for r in rows[:-1]])
    toff = np.cumsum([0] + [r["targets_owned"] for r in rows[:-1]])
    for r, so, to in zip(rows, soff, toff):
        assert r["nboxes_global"] == h.nboxes and r["nlevels_global"] == h.nlevels
        assert r["nsources_global"] == nranks * n and r["ntargets_global"] == nranks * nt
        assert r["source_offset"] == int(so) and r["target_offset"] == int(to)
        assert r["let_nboxes"] <= h.nboxes
    deep = np.nonzero(h.box_levels > top_level)[0].astype(np.uint64)
    assert sum(r["deep_ids"] for r in rows) == int(deep.sum())
    # particle identity from C: sum over tree positions p of (p + 1) * (global user id of the source
    # at p), over all ranks, is the single-GPU tree's (mod 2^64) -- every rank's ids, in its slice
    want = int((np.arange(1, h.nsources + 1, dtype=np.uint64) * h.user_source_ids.astype(np.uint64)).sum())
    assert sum(r["user_id_digest"] for r in rows) % 2**64 == want
    assert [r["chunk_offset"] for r in rows] == [k * n for k in range(nranks)]
