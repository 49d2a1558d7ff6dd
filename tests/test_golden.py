"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the
validated oracle) and a micro-case derived by hand from the reference's source
lines.  CPU: the oracle still reproduces them.  GPU: the device output equals them."""

import glob
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg  # noqa: E402

NAMES = sorted(os.path.splitext(os.path.basename(f))[0]
               for f in glob.glob(os.path.join(HERE, "golden", "*.npz"))
               if os.path.basename(f) != "reference_vectors.npz")   # test_reference_vectors.py


def compare_with_fixture(name, inp, tree, trav):
    want = np.load(os.path.join(HERE, "golden", name + ".npz"))
    got = mg.flatten(inp, tree, trav)
    assert set(got) == set(want.files), set(got) ^ set(want.files)
    for key in want.files:
        a, b = np.asarray(got[key]), want[key]
        assert a.dtype == b.dtype and a.shape == b.shape, key
        assert np.array_equal(a, b), key


def test_fixture_list_matches_generator():
    assert NAMES == sorted(mg.CASES)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(oracle, name):
    inp, tree, trav = mg.build(oracle, mg.CASES[name])
    compare_with_fixture(name, inp, tree, trav)


# {{{ hand-derived micro-case: 2D, 3x3 lattice, max_particles_in_box=1

def lattice_case():
    """Points (i/2, j/2), i, j in 0..2, listed row by row (x varies slowest).

    By hand from the cited lines: bbox = [0,1]^2, root_extent = 1.0001
    (tree_build.py:464-476), so the level-1 split is at 0.50005 and coordinates 0 and
    0.5 fall in the low half, 1 in the high half; Morton number = 2*xbit + ybit
    (tbk:441-445).  Level 1: box 1 (lo,lo) holds the 4 points with x,y in {0,.5},
    box 2 (lo,hi) = {(0,1),(.5,1)}, box 3 (hi,lo) = {(1,0),(1,.5)}, box 4 = {(1,1)}.
    With max_particles_in_box=1 boxes 1-3 split (tbk:577-591); their children split
    at 0.250025 / 0.750075: box 1 -> 4 single-point boxes 5..8; box 2 -> (lo,hi),
    (hi,hi) = boxes 9, 10; box 3 -> (hi,lo), (hi,hi) = boxes 11, 12 (empty children
    pruned, tbk:1707-1716).  13 boxes, 3 levels."""
    x = np.repeat([0.0, 0.5, 1.0], 3)
    y = np.tile([0.0, 0.5, 1.0], 3)
    #   id: 0:(0,0) 1:(0,.5) 2:(0,1) 3:(.5,0) 4:(.5,.5) 5:(.5,1) 6:(1,0) 7:(1,.5) 8:(1,1)
    ext = np.float64(1.0) * (1 + 1e-4)
    c = lambda k, lev: (k + 0.5) * ext / 2 ** lev       # noqa: E731  centre of cell k
    expect = dict(
        nboxes=13, nlevels=3, level_start_box_nrs=[0, 1, 5, 13],
        box_levels=[0, 1, 1, 1, 1] + [2] * 8,
        box_parent_ids=[0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3],
        # rows = Morton number of the child, columns = boxes 0..12
        box_child_ids=[[1, 5, 0, 0, 0] + [0] * 8,
                       [2, 6, 9, 0, 0] + [0] * 8,
                       [3, 7, 0, 11, 0] + [0] * 8,
                       [4, 8, 10, 12, 0] + [0] * 8],
        # tree order: box 1's children (0,0),(0,.5),(.5,0),(.5,.5); box 2's; box 3's; box 4
        user_source_ids=[0, 1, 3, 4, 2, 5, 6, 7, 8],
        box_source_starts=[0, 0, 4, 6, 8, 0, 1, 2, 3, 4, 5, 6, 7],
        box_source_counts_cumul=[9, 4, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1],
        box_centers_x=[c(0, 0), c(0, 1), c(0, 1), c(1, 1), c(1, 1),
                       c(0, 2), c(0, 2), c(1, 2), c(1, 2), c(0, 2), c(1, 2), c(3, 2), c(3, 2)],
        box_centers_y=[c(0, 0), c(0, 1), c(1, 1), c(0, 1), c(1, 1),
                       c(0, 2), c(1, 2), c(0, 2), c(1, 2), c(3, 2), c(3, 2), c(0, 2), c(1, 2)],
        # leaves: 1 = IS_SOURCE|IS_TARGET (3); parents: HAS_*_CHILD (12), tree.py:109-145
        box_flags=[12, 12, 12, 12, 3] + [3] * 8,
    )
    return [x, y], expect


def check_lattice(tree):
    _, e = lattice_case()
    assert tree.nboxes == e["nboxes"] and tree.nlevels == e["nlevels"]
    nb = tree.nboxes
    assert tree.level_start_box_nrs.tolist() == e["level_start_box_nrs"]
    assert tree.box_levels.tolist() == e["box_levels"]
    assert tree.box_parent_ids.tolist() == e["box_parent_ids"]
    assert tree.box_child_ids[:, :nb].tolist() == e["box_child_ids"]
    assert tree.user_source_ids.tolist() == e["user_source_ids"]
    assert tree.box_source_starts.tolist() == e["box_source_starts"]
    assert tree.box_source_counts_cumul.tolist() == e["box_source_counts_cumul"]
    assert tree.box_flags.tolist() == e["box_flags"]
    assert np.allclose(tree.box_centers[0, :nb], e["box_centers_x"], rtol=1e-15, atol=0)
    assert np.allclose(tree.box_centers[1, :nb], e["box_centers_y"], rtol=1e-15, atol=0)
    assert float(tree.root_extent) == 1.0 * (1 + 1e-4)


def test_oracle_hand_derived_lattice(oracle):
    pts, _ = lattice_case()
    check_lattice(oracle.build_tree(pts, max_particles_in_box=1))

# }}}


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_device_reproduces_golden(name):
    from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder
    actx = HIPArrayContext(0)
    case = mg.CASES[name]
    inp = mg.make_inputs(case)
    kw = dict(case["kw"])
    if inp["targets"] is not None:
        kw["targets"] = [actx.from_numpy(t) for t in inp["targets"]]
    if inp["target_radii"] is not None:
        kw["target_radii"] = actx.from_numpy(inp["target_radii"])
    tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(p) for p in inp["particles"]], **kw)
    trav, _ = FMMTraversalBuilder(actx, **case.get("trav_kw", {}))(actx, tree)
    compare_with_fixture(name, inp, actx.to_numpy(tree), actx.to_numpy(trav))


@pytest.mark.gpu
def test_device_hand_derived_lattice():
    from boxtree_amd import HIPArrayContext, TreeBuilder
    actx = HIPArrayContext(0)
    pts, _ = lattice_case()
    tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(p) for p in pts], max_particles_in_box=1)
    check_lattice(actx.to_numpy(tree))


@pytest.mark.gpu
def test_device_coincident_points_raise():
    """SURVEY 8c error case: coincident points beyond the leaf capacity."""
    from boxtree_amd import HIPArrayContext, MaxLevelsExceeded, TreeBuilder
    actx = HIPArrayContext(0)
    pts = [actx.from_numpy(np.full(100, 0.25)) for _ in range(2)]
    pts[0][50:] = 0.75
    with pytest.raises(MaxLevelsExceeded):
        TreeBuilder(actx)(actx, pts, max_particles_in_box=10)
