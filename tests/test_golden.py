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


def lattice_traversal():
    """Every interaction list of the lattice tree above, IN ORDER, worked out by hand
    from boxtree/traversal.py:398-1146 (well_sep_is_n_away = 1, no extents).

    Level-2 boxes as cells (ix, iy) of the 4x4 grid: 5 (0,0) 6 (0,1) 7 (1,0) 8 (1,1)
    9 (0,3) 10 (1,3) 11 (3,0) 12 (3,1); box 4 is the level-1 leaf over cells 2..3 x 2..3.
    Two boxes are adjacent iff their cell ranges touch or overlap on every axis
    (:255-320).  The walks visit children in Morton order, i.e. the boxes in the order
    1 (5 6 7 8) 2 (9 10) 3 (11 12) 4.

    * colleagues (:398-464): same level, adjacent, self excluded; the root has none.
    * list 1 (:470-550), per target box = leaf: every adjacent leaf, itself included,
      in walk order.  Box 4 touches 8 (corner), 10 and 12 (edges); 8, 10, 12 in turn see 4.
    * list 2 (:556-601): children of the parent's colleagues that are not adjacent.
    * list 3 (:607-875): below the colleagues of a target box, the first boxes that are no
      longer adjacent.  Only box 4 has colleagues with children: of 1's children 5, 6, 7
      are separated (8 touches), of 2's 9 (10 touches), of 3's 11 (12 touches).
    * list 4 (:931-1146): for a level-2 box, the leaf colleagues of its parent (only box 4
      is a level-1 leaf) that it does not touch itself: 5, 6, 7, 9, 11 get [4]; 8, 10, 12
      touch 4."""
    leaves = [4, 5, 6, 7, 8, 9, 10, 11, 12]
    coll = {0: [], 1: [2, 3, 4], 2: [1, 3, 4], 3: [1, 2, 4], 4: [1, 2, 3],
            5: [6, 7, 8], 6: [5, 7, 8], 7: [5, 6, 8], 8: [5, 6, 7],
            9: [10], 10: [9], 11: [12], 12: [11]}
    list1 = {4: [8, 10, 12, 4], 5: [5, 6, 7, 8], 6: [5, 6, 7, 8], 7: [5, 6, 7, 8],
             8: [5, 6, 7, 8, 4], 9: [9, 10], 10: [9, 10, 4], 11: [11, 12], 12: [11, 12, 4]}
    list2 = {b: [] for b in range(5)}
    list2.update({b: [9, 10, 11, 12] for b in (5, 6, 7, 8)})
    list2.update({b: [5, 6, 7, 8, 11, 12] for b in (9, 10)})
    list2.update({b: [5, 6, 7, 8, 9, 10] for b in (11, 12)})
    list3_level2 = {4: [5, 6, 7, 9, 11]}           # target box -> list; source level 2 only
    list4 = {b: [] for b in range(13)}
    list4.update({b: [4] for b in (5, 6, 7, 9, 11)})
    return dict(leaves=leaves, coll=coll, list1=list1, list2=list2, list3_level2=list3_level2,
                list4=list4)


def check_lattice_traversal(trav):
    e = lattice_traversal()

    def rows(starts, lists):
        starts, lists = np.asarray(starts), np.asarray(lists)
        return [lists[starts[i]:starts[i + 1]].tolist() for i in range(len(starts) - 1)]

    assert np.asarray(trav.source_boxes).tolist() == e["leaves"]
    assert np.asarray(trav.target_boxes).tolist() == e["leaves"]
    assert np.asarray(trav.source_parent_boxes).tolist() == [0, 1, 2, 3]
    assert np.asarray(trav.target_or_target_parent_boxes).tolist() == list(range(13))
    # first list entry of every level (:361-392, 2093-2096)
    assert np.asarray(trav.level_start_source_box_nrs).tolist() == [0, 0, 1, 9]
    assert np.asarray(trav.level_start_target_box_nrs).tolist() == [0, 0, 1, 9]
    assert np.asarray(trav.level_start_source_parent_box_nrs).tolist() == [0, 1, 4, 4]
    assert np.asarray(trav.level_start_target_or_target_parent_box_nrs).tolist() == [0, 1, 5, 13]
    assert rows(trav.same_level_non_well_sep_boxes_starts,
                trav.same_level_non_well_sep_boxes_lists) == [e["coll"][b] for b in range(13)]
    assert rows(trav.neighbor_source_boxes_starts,
                trav.neighbor_source_boxes_lists) == [e["list1"][b] for b in e["leaves"]]
    assert rows(trav.from_sep_siblings_starts,
                trav.from_sep_siblings_lists) == [e["list2"][b] for b in range(13)]
    assert rows(trav.from_sep_bigger_starts,
                trav.from_sep_bigger_lists) == [e["list4"][b] for b in range(13)]
    assert trav.from_sep_close_smaller_starts is None and trav.from_sep_close_bigger_starts is None
    # list 3: one BuiltList per source level, empty lists eliminated (:2211-2215)
    l3 = trav.from_sep_smaller_by_level
    assert len(l3) == 3
    for lev in (0, 1):
        assert l3[lev].count == 0 and l3[lev].num_nonempty_lists == 0
        assert np.asarray(l3[lev].starts).tolist() == [0]
        assert np.asarray(l3[lev].compressed_indices).tolist() == [0] * 10
        assert np.asarray(trav.target_boxes_sep_smaller_by_source_level[lev]).tolist() == []
    assert l3[2].count == 5 and l3[2].num_nonempty_lists == 1
    assert np.asarray(l3[2].starts).tolist() == [0, 5]
    assert np.asarray(l3[2].lists).tolist() == e["list3_level2"][4]
    assert np.asarray(l3[2].nonempty_indices).tolist() == [0]       # target box number of box 4
    assert np.asarray(l3[2].compressed_indices).tolist() == [0] + [1] * 9
    assert np.asarray(trav.target_boxes_sep_smaller_by_source_level[2]).tolist() == [4]


def test_oracle_hand_derived_lattice_traversal(oracle):
    pts, _ = lattice_case()
    tree = oracle.build_tree(pts, max_particles_in_box=1)
    check_lattice_traversal(oracle.build_traversal(tree))


def stuck_target_case():
    """One target with an extent that gets stuck (tbk:388-403), derived by hand.

    Sources s0 = (0, 0), s1 = (.45, .45), s2 = (1, 1); one target t = (.3, .3) with
    radius .2; stick_out_factor .25, linf, max_particles_in_box = 1.  The bounding box of
    x -+ r is [0, 1]^2, root_extent = 1.0001.  Level 1: t's cell is (0, 0) with centre
    .250025 and stick-out radius (1 + .25) / 2 * 1.0001 / 2 = .31253125; .3 + .2 >= .5626
    and .3 - .2 < -.0625 are both false, so t descends with s0 and s1 into box 1 = (lo, lo);
    s2 is alone in box 2 = (hi, hi) (empty boxes are pruned).  Level 2: t's cell is
    (1, 1) with centre .3750375 and stick-out radius .156265625; .3 - .2 < .21877 holds:
    t stops in box 1 (Morton number -1, tbk:448-451).  Box 1 still splits, because its
    child-bound weight is 2 (tbk:568-573): s0 -> box 3 = cell (0, 0), s1 -> box 4 = cell
    (1, 1).  In tree order a box's own particles come first (tbk:163): t, s0, s1, s2."""
    src = [np.array([0.0, 0.45, 1.0]), np.array([0.0, 0.45, 1.0])]
    tgt = [np.array([0.3]), np.array([0.3])]
    kw = dict(target_radii=np.array([0.2]), stick_out_factor=0.25, max_particles_in_box=1)
    ext = np.float64(1.0) * (1 + 1e-4)
    c = lambda k, lev: (k + 0.5) * ext / 2 ** lev       # noqa: E731
    expect = dict(
        nboxes=5, nlevels=3, level_start_box_nrs=[0, 1, 3, 5],
        box_levels=[0, 1, 1, 2, 2], box_parent_ids=[0, 0, 0, 1, 1],
        box_child_ids=[[1, 3, 0, 0, 0], [0] * 5, [0] * 5, [2, 4, 0, 0, 0]],
        user_source_ids=[0, 1, 2], sorted_target_ids=[0],
        box_source_starts=[0, 0, 2, 0, 1], box_source_counts_cumul=[3, 2, 1, 1, 1],
        box_source_counts_nonchild=[0, 0, 1, 1, 1],
        # the stuck target: box 1 has one target of its own; targets before a box's range
        box_target_starts=[0, 0, 1, 1, 1], box_target_counts_cumul=[1, 1, 0, 0, 0],
        box_target_counts_nonchild=[0, 1, 0, 0, 0],
        # 12 = HAS_SOURCE|TARGET_CHILD_BOXES on every parent (tbk:1252-1256), +2 = IS_TARGET_BOX
        box_flags=[12, 14, 1, 1, 1],
        box_centers=[c(0, 0), c(0, 1), c(1, 1), c(0, 2), c(1, 2)],
        # target extents (tbk:1311-1399): own targets -+ radius, the children's boxes, and
        # the box centre to start from (an empty box keeps its centre)
        tgt_bbox_min=[0.3 - 0.2, 0.3 - 0.2, c(1, 1), c(0, 2), c(1, 2)],
        tgt_bbox_max=[c(1, 1), 0.3 + 0.2, c(1, 1), c(0, 2), c(1, 2)],
    )
    return src, tgt, kw, expect


def check_stuck_target(tree):
    _, _, _, e = stuck_target_case()
    nb = tree.nboxes
    assert nb == e["nboxes"] and tree.nlevels == e["nlevels"]
    for name in ("level_start_box_nrs", "box_levels", "box_parent_ids", "user_source_ids",
                 "sorted_target_ids", "box_source_starts", "box_source_counts_cumul",
                 "box_source_counts_nonchild", "box_target_starts", "box_target_counts_cumul",
                 "box_target_counts_nonchild", "box_flags"):
        assert np.asarray(getattr(tree, name)).tolist() == e[name], name
    assert tree.box_child_ids[:, :nb].tolist() == e["box_child_ids"]
    for ax in range(2):
        assert np.allclose(tree.box_centers[ax, :nb], e["box_centers"], rtol=1e-15, atol=0)
        assert np.allclose(tree.box_target_bounding_box_min[ax, :nb], e["tgt_bbox_min"], rtol=1e-15, atol=0)
        assert np.allclose(tree.box_target_bounding_box_max[ax, :nb], e["tgt_bbox_max"], rtol=1e-15, atol=0)
    # particles in tree order
    assert np.asarray(tree.targets[0]).tolist() == [0.3] and np.asarray(tree.target_radii).tolist() == [0.2]
    assert np.asarray(tree.sources[0]).tolist() == [0.0, 0.45, 1.0]


def test_oracle_hand_derived_stuck_target(oracle):
    src, tgt, kw, _ = stuck_target_case()
    check_stuck_target(oracle.build_tree(src, targets=tgt, **kw))

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
@pytest.mark.parametrize("force_generic", [False, True])
def test_device_hand_derived_lattice_traversal(force_generic):
    from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder
    actx = HIPArrayContext(0)
    pts, _ = lattice_case()
    tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(p) for p in pts], max_particles_in_box=1)
    trav, _ = FMMTraversalBuilder(actx)(actx, tree, _force_generic=force_generic)
    check_lattice_traversal(actx.to_numpy(trav))


@pytest.mark.gpu
def test_device_hand_derived_stuck_target():
    from boxtree_amd import HIPArrayContext, TreeBuilder
    actx = HIPArrayContext(0)
    src, tgt, kw, _ = stuck_target_case()
    dkw = dict(kw, target_radii=actx.from_numpy(kw["target_radii"]))
    tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(p) for p in src],
                                targets=[actx.from_numpy(p) for p in tgt], **dkw)
    check_stuck_target(actx.to_numpy(tree))


@pytest.mark.gpu
def test_device_coincident_points_raise():
    """SURVEY 8c error case: coincident points beyond the leaf capacity."""
    from boxtree_amd import HIPArrayContext, MaxLevelsExceeded, TreeBuilder
    actx = HIPArrayContext(0)
    pts = [actx.from_numpy(np.full(100, 0.25)) for _ in range(2)]
    pts[0][50:] = 0.75
    with pytest.raises(MaxLevelsExceeded):
        TreeBuilder(actx)(actx, pts, max_particles_in_box=10)
