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
