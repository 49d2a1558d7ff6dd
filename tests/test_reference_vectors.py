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
