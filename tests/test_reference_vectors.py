"""tests/golden/reference_vectors.npz holds outputs of the reference itself: the
plain Python / numpy parts of boxtree, executed in the build container by
tests/golden/make_reference_vectors.py (see its docstring for what was run and how).

CPU: the restatements (oracle/oracle.py, tests/invariants.py) and the host-side
product code reproduce those outputs.  GPU: the device FMM stages, run on the
device-built trees of the golden cases, reproduce what the reference's own
``drive_fmm`` + ``ConstantOneExpansionWrangler`` computed on the oracle's trees."""

import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg  # noqa: E402

VEC = np.load(os.path.join(HERE, "golden", "reference_vectors.npz"))
STAGES = ("form_multipoles", "coarsen_multipoles", "eval_direct", "multipole_to_local",
          "eval_multipoles", "form_locals", "refine_locals", "eval_locals")


def test_generator_runs_where_the_reference_is():
    """The vectors can be regenerated (only) where /root/reference exists."""
    if not os.path.isdir("/root/reference"):
        pytest.skip("reference checkout not present (GPU box)")
    import make_reference_vectors as mrv
    out = {}
    mrv.comm_pattern_vectors(out)
    mrv.class_vectors(out)
    for k, v in out.items():
        assert np.array_equal(VEC[k], v), k


def test_comm_pattern_matches_reference():
    from boxtree_amd.distributed.calculation import (reduce_scatter_num_stages,
                                                     reduce_scatter_stage)
    rows = VEC["comm_pattern/rows"]
    state = {}
    nstages = {}
    for size, rank, stage, sink, nsrc, s0, s1, lo, hi in rows.tolist():
        left, right = state.get((size, rank), (0, size))
        sinks, sources, users, nxt = reduce_scatter_stage(rank, left, right)
        assert sinks == [sink], (size, rank, stage)
        assert sorted(sources) == sorted(s for s in (s0, s1)[:nsrc]), (size, rank, stage)
        assert users == (lo, hi), (size, rank, stage)
        state[(size, rank)] = nxt
        nstages[size] = max(nstages.get(size, 0), stage + 1)
    # every rank of the reference pattern ends on itself exactly when ours does
    for (size, rank), (left, right) in state.items():
        assert (left, right) == (rank, rank + 1)
    for size, n in nstages.items():
        assert reduce_scatter_num_stages(size) == n


def test_dfs_order_and_partition_match_reference(oracle):
    for idx in range(int(VEC["partition/ncases"])):
        pre = f"partition/{idx}/"
        child = VEC[pre + "box_child_ids"]
        tree = SimpleNamespace(nboxes=child.shape[1], dimensions=int(VEC[pre + "dims"]),
                               box_child_ids=child)
        order = oracle.dfs_order(tree)
        assert np.array_equal(order, VEC[pre + "dfs_order"])
        for key in VEC.files:
            if not key.startswith(pre + "segments/"):
                continue
            _, _, _, cname, size = key.split("/")
            size = int(size)
            cost = VEC[pre + "cost/" + cname]
            want = VEC[key]                    # the rows the reference's loop wrote
            got = oracle.partition_work_segments(cost, order, size)
            assert np.array_equal(got[:len(want)], want), key
            # rows the loop never reaches are uninitialised upstream; ours are empty
            assert np.all(got[len(want):] == tree.nboxes), key


def test_class_arithmetic_matches_reference():
    from boxtree_amd.rotation_classes import RotationClassesBuilder
    rcb = RotationClassesBuilder(None)
    for nway in (1, 2, 3):
        for dims in (2, 3):
            pre = f"classes/{nway}_{dims}/"
            vecs = VEC[pre + "vectors"]
            assert rcb.tcb.ntranslation_classes_per_level(nway, dims) == len(vecs)
            for cls in range(len(vecs)):
                assert np.array_equal(
                    rcb.tcb.translation_class_to_normalized_vector(nway, dims, cls), vecs[cls])
            for label in ("all", "some"):
                to_rot, angles = rcb.compute_rotation_classes(nway, dims,
                                                              VEC[pre + label + "/used"])
                assert np.array_equal(to_rot, VEC[pre + label + "/to_rot_class"])
                # float64 angles bit for bit (same arccos of the same reduced vector)
                assert np.array_equal(np.array(angles), VEC[pre + label + "/angles"])


def _fmm_keys(name, label):
    pre = f"fmm/{name}/{label}/"
    return pre, sorted(k[len(pre):] for k in VEC.files if k.startswith(pre))


@pytest.mark.parametrize("name", sorted(mg.CASES))
def test_fmm_stages_of_the_restatement_match_reference(oracle, name):
    """tests/invariants.py's statement of drive_fmm + constant-one wrangler, which the
    full-size GPU completeness tests are checked against, equals the reference's run
    stage by stage (on the oracle's tree and lists, which is what the reference ran on)."""
    from invariants import constant_one_stages
    _inp, tree, trav = mg.build(oracle, mg.CASES[name])
    for label in ("ones", "rand"):
        pre, keys = _fmm_keys(name, label)
        got = constant_one_stages(tree, trav, VEC[pre + "weights"])
        assert sorted(got) == keys, set(got) ^ set(keys)
        for k in keys:
            assert np.array_equal(np.asarray(got[k], np.float64), VEC[pre + k]), (label, k)
        total = VEC[pre + "weights"].sum()
        assert np.all(VEC[pre + "potentials"] == total)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(mg.CASES))
def test_device_fmm_stages_match_reference(name):
    import torch
    from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder
    from boxtree_amd.constant_one import (ConstantOneExpansionWrangler,
                                          ConstantOneTreeIndependentDataForWrangler)
    from boxtree_amd.fmm import drive_fmm
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

    class Recording(ConstantOneExpansionWrangler):
        def __init__(self, *args):
            super().__init__(*args)
            self.stages = {}

    def record(stage):
        base = getattr(ConstantOneExpansionWrangler, stage)

        def method(self, *args, **kwargs):
            result = base(self, *args, **kwargs)
            n = sum(k.startswith(stage) for k in self.stages)
            self.stages[f"{stage}_{n}"] = result.detach().cpu().numpy().copy()
            return result
        return method

    for stage in STAGES:
        setattr(Recording, stage, record(stage))

    for label in ("ones", "rand"):
        pre, keys = _fmm_keys(name, label)
        weights = torch.from_numpy(VEC[pre + "weights"]).cuda()
        wrangler = Recording(ConstantOneTreeIndependentDataForWrangler(), trav)
        pot = drive_fmm(actx, wrangler, [weights])
        got = dict(wrangler.stages, weights=VEC[pre + "weights"],
                   potentials=pot.cpu().numpy())
        assert sorted(got) == keys, set(got) ^ set(keys)
        for k in keys:
            assert np.array_equal(got[k], VEC[pre + k]), (label, k)


@pytest.mark.gpu
def test_device_dfs_order_and_partition_match_reference():
    import ctypes as ct
    from boxtree_amd import HIPArrayContext
    from boxtree_amd.array_context import ptr
    actx = HIPArrayContext(0)
    for idx in range(int(VEC["partition/ncases"])):
        pre = f"partition/{idx}/"
        child = np.ascontiguousarray(VEC[pre + "box_child_ids"])
        nchildren, nboxes = child.shape
        # the oracle numbers boxes level by level: level starts from a sweep over the links
        level = np.zeros(nboxes, np.int64)
        for b in range(nboxes):
            ch = child[:, b]
            level[ch[ch > 0]] = level[b] + 1
        assert np.all(np.diff(level) >= 0)
        lsb = np.concatenate([[0], np.cumsum(np.bincount(level))]).astype(np.int32)
        order = actx.empty(nboxes, np.int32)
        assert actx.lib.bt_dfs_order(
            actx.handle, nchildren, len(lsb) - 1, lsb.ctypes.data_as(ct.POINTER(ct.c_int32)),
            nboxes, nboxes, ptr(actx.from_numpy(child)), ptr(order)) == 0
        assert np.array_equal(order.cpu().numpy(), VEC[pre + "dfs_order"])
        for key in VEC.files:
            if not key.startswith(pre + "segments/"):
                continue
            _, _, _, cname, size = key.split("/")
            size = int(size)
            if cname == "float":
                continue      # not exactly summable: the device prefix sum may round
                              # differently from the reference's running sum
            seg = np.zeros((size, 2), np.int32)
            assert actx.lib.bt_partition_work(
                actx.handle, nboxes, ptr(order), ptr(actx.from_numpy(VEC[pre + "cost/" + cname])),
                size, seg.ctypes.data_as(ct.POINTER(ct.c_int32))) == 0
            want = VEC[key]
            assert np.array_equal(seg[:len(want)], want), key
            assert np.all(seg[len(want):] == nboxes), key


COST_KEYS = ("process_form_multipoles", "get_ndirect_sources_per_target_box", "process_direct",
             "process_list2", "process_list3", "process_list4", "process_eval_locals",
             "process_coarsen_multipoles", "process_refine_locals")


@pytest.mark.parametrize("name", sorted(mg.CASES))
def test_cost_model_loops_match_reference(oracle, name):
    """oracle.cost_model's per-stage pieces == the reference's _PythonFMMCostModel loops
    (run on the same oracle tree with the same per-level factors)."""
    import make_reference_vectors as mrv
    _inp, tree, trav = mg.build(oracle, mg.CASES[name])
    tc = mrv.cost_factors(mg.CASES[name]["seed"], tree.nlevels)
    _, _, pieces = oracle.cost_model_from_factors(tree, trav, tc)
    for k in COST_KEYS:
        assert np.array_equal(np.asarray(pieces[k], np.float64), VEC[f"cost/{name}/{k}"]), k


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(mg.CASES))
def test_device_cost_model_loops_match_reference(name):
    import make_reference_vectors as mrv
    from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder
    from boxtree_amd.cost import FMMCostModel
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
    f = {k: (actx.from_numpy(v) if np.ndim(v) else float(v))
         for k, v in mrv.cost_factors(case["seed"], int(tree.nlevels)).items()}
    m = FMMCostModel()
    nd = m.get_ndirect_sources_per_target_box(actx, trav)
    got = {
        "process_form_multipoles": m.process_form_multipoles(actx, trav, f["p2m_cost"]),
        "get_ndirect_sources_per_target_box": nd,
        "process_direct": m.process_direct(actx, trav, nd, f["c_p2p"]),
        "process_list2": m.process_list2(actx, trav, f["m2l_cost"]),
        "process_list3": m.process_list3(actx, trav, f["m2p_cost"]),
        "process_list4": m.process_list4(actx, trav, f["p2l_cost"]),
        "process_eval_locals": m.process_eval_locals(actx, trav, f["l2p_cost"]),
        "process_coarsen_multipoles": m.process_coarsen_multipoles(actx, trav, f["m2m_cost"]),
        "process_refine_locals": m.process_refine_locals(actx, trav, f["l2l_cost"]),
    }
    for k in COST_KEYS:
        v = got[k]
        v = v.cpu().numpy() if hasattr(v, "cpu") else np.float64(v)
        assert np.array_equal(np.asarray(v, np.float64), VEC[f"cost/{name}/{k}"]), k


# {{{ pure-box trees made by the reference's boxtree.tree_of_boxes

TOB_NAMES = [str(n) for n in VEC["tob/names"]]
TOB_TRAV_FIELDS = [f for f in mg.TRAV_FIELDS if "close" not in f]


def _reference_tob(name):
    """The tree exactly as the reference's functions return it (int32 levels, root
    parent -1, no level starts, [2^d, nboxes] child ids, IS_LEAF_BOX flags)."""
    from boxtree_amd.tree import TreeOfBoxes
    pre = f"tob/{name}/"
    a = {k[len(pre):]: VEC[k] for k in VEC.files if k.startswith(pre)}
    return TreeOfBoxes(
        root_extent=a["root_extent"][()], box_centers=a["box_centers"],
        box_parent_ids=a["box_parent_ids"], box_child_ids=a["box_child_ids"],
        box_levels=a["box_levels"], box_flags=a["box_flags"], level_start_box_nrs=None,
        box_id_dtype=np.dtype(np.int32), box_level_dtype=np.dtype(np.int32),
        coord_dtype=a["box_centers"].dtype, sources_have_extent=False,
        targets_have_extent=False, extent_norm="linf", stick_out_factor=0, _is_pruned=True)


def _oracle_view(tob):
    levels = np.asarray(tob.box_levels)
    nlevels = int(levels.max()) + 1
    starts = np.concatenate([[0], np.cumsum(np.bincount(levels, minlength=nlevels))])
    parents = np.array(tob.box_parent_ids, np.int32)
    parents[0] = 0                              # a Tree's root is its own parent
    return SimpleNamespace(
        coord_dtype=np.dtype(tob.coord_dtype), dimensions=tob.box_centers.shape[0],
        nboxes=len(levels), aligned_nboxes=len(levels), nlevels=nlevels,
        root_extent=tob.root_extent, box_centers=np.ascontiguousarray(tob.box_centers),
        box_levels=levels.astype(np.uint8), box_child_ids=np.ascontiguousarray(
            tob.box_child_ids, dtype=np.int32),
        box_flags=np.asarray(tob.box_flags, np.uint8), box_parent_ids=parents,
        level_start_box_nrs=starts.astype(np.int32), sources_are_targets=True,
        sources_have_extent=False, targets_have_extent=False, extent_norm="linf",
        stick_out_factor=0, _is_pruned=True)


@pytest.mark.parametrize("name", TOB_NAMES)
def test_oracle_traversal_of_reference_made_box_trees(oracle, name):
    """Structure of the fixtures + the oracle's lists on them: every box is source and
    target, so list 1 of a box must hold the box itself, colleagues are symmetric,
    and list 2 entries sit on the same level, non-adjacent, under adjacent parents."""
    tob = _reference_tob(name)
    t = _oracle_view(tob)
    assert np.all(np.diff(t.box_levels.astype(int)) >= 0)
    for b in range(1, t.nboxes):
        assert b in tob.box_child_ids[:, tob.box_parent_ids[b]]
    trav = oracle.build_traversal(t)
    assert np.array_equal(trav.source_boxes, np.arange(t.nboxes))
    st, li = trav.neighbor_source_boxes_starts, trav.neighbor_source_boxes_lists
    for i, b in enumerate(trav.target_boxes):
        assert b in li[st[i]:st[i + 1]]
    st, li = trav.same_level_non_well_sep_boxes_starts, trav.same_level_non_well_sep_boxes_lists
    coll = [set(li[st[b]:st[b + 1]]) for b in range(t.nboxes)]
    for b in range(t.nboxes):
        for c in coll[b]:
            assert b in coll[c] and t.box_levels[c] == t.box_levels[b]
    st, li = trav.from_sep_siblings_starts, trav.from_sep_siblings_lists
    for i, b in enumerate(trav.target_or_target_parent_boxes):
        for s in li[st[i]:st[i + 1]]:
            assert t.box_levels[s] == t.box_levels[b] and s not in coll[b]
            assert t.box_parent_ids[s] in coll[t.box_parent_ids[b]]


@pytest.mark.gpu
@pytest.mark.parametrize("name", TOB_NAMES)
def test_device_traversal_of_reference_made_box_trees(oracle, name):
    """FMMTraversalBuilder takes the reference's TreeOfBoxes as is (numpy arrays,
    test/test_tree_of_boxes.py:240-270) and gives the oracle's lists."""
    from boxtree_amd import FMMTraversalBuilder, HIPArrayContext
    actx = HIPArrayContext(0)
    tob = _reference_tob(name)
    want = oracle.build_traversal(_oracle_view(tob))
    got = actx.to_numpy(FMMTraversalBuilder(actx)(actx, tob)[0])
    for f in TOB_TRAV_FIELDS:
        assert np.array_equal(getattr(got, f), getattr(want, f)), f
    for lev, bl in enumerate(want.from_sep_smaller_by_level):
        g = got.from_sep_smaller_by_level[lev]
        assert np.array_equal(g.starts, bl.starts) and np.array_equal(g.lists, bl.lists)
        assert np.array_equal(got.target_boxes_sep_smaller_by_source_level[lev],
                              want.target_boxes_sep_smaller_by_source_level[lev])

# }}}


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["adaptive_2d", "adaptive_3d_coarsened"])
def test_device_peer_lists_of_reference_made_box_trees(oracle, name):
    from boxtree_amd import HIPArrayContext
    from boxtree_amd.area_query import PeerListFinder
    actx = HIPArrayContext(0)
    tob = _reference_tob(name)
    view = _oracle_view(tob)
    lows = view.box_centers[:, 0] - 0.5 * view.root_extent
    view.bounding_box = (lows, lows + view.root_extent)
    want = oracle.peer_lists(view)
    got, _ = PeerListFinder(actx)(actx, tob)
    got = actx.to_numpy(got)
    assert np.array_equal(got.peer_list_starts, want.peer_list_starts)
    assert np.array_equal(got.peer_lists, want.peer_lists)
