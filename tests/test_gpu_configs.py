"""BASELINE.json's configurations in the -m gpu suite (SURVEY.md section 8d recipes).

Every recipe is built on the device and compared, array by array, with the CPU
oracle at a size the oracle finishes in seconds (c1 at its exact size); the big
configurations are additionally run at full size through the reference's
interaction-completeness test (test/test_fmm.py:141-391: constant-one kernel,
every target must see every source exactly once).

  c1  2D uniform, default_rng(15).random, 10^5 points, mpb 30   exact recipe vs oracle
  c2  3D uniform, 10^7, mpb 64                                  2*10^6 vs oracle (+ full size
                                                                in test_gpu_fmm.py)
  c3  3D sphere surface 10^8, mpb 64, and its clustered variant 2*10^6 vs oracle, full size
                                                                completeness (c3: test_gpu_fmm.py,
                                                                c3c: here)
  c4  10^8 sources + 10^7 targets with radii                    10^6 + 10^5 vs oracle, full size
                                                                completeness here
  c5  10^9 over 8 GPUs                                          needs 8 GPUs; the sharded path is
                                                                covered by test_gpu_parity.py
"""

import numpy as np
import pytest

from compare import assert_same_traversal, assert_same_tree
from invariants import check_traversal, check_tree

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def actx():
    from boxtree_amd import HIPArrayContext
    return HIPArrayContext(0)


def recipe(workload, n):
    from bench import WORKLOAD_MPB, make_workload_numpy
    w = make_workload_numpy(workload, n, 15)
    return w, WORKLOAD_MPB.get(workload, 64)


def build_and_compare(actx, oracle, w, mpb, both_paths=True):
    from boxtree_amd import FMMTraversalBuilder, TreeBuilder
    dev = lambda arrs: None if arrs is None else [actx.from_numpy(a) for a in arrs]  # noqa: E731
    kw = dict(w["kw"])
    dkw = dict(kw)
    if "target_radii" in dkw:
        dkw["target_radii"] = actx.from_numpy(dkw["target_radii"])
    tree, _ = TreeBuilder(actx)(actx, dev(w["particles"]), targets=dev(w["targets"]),
                                max_particles_in_box=mpb, **dkw)
    otree = oracle.build_tree(w["particles"], targets=w["targets"], max_particles_in_box=mpb,
                              **kw)
    htree = actx.to_numpy(tree)
    assert_same_tree(htree, otree)
    otrav = oracle.build_traversal(otree)
    htrav = None
    for force_generic in ((True, False) if both_paths else (False,)):
        trav, _ = FMMTraversalBuilder(actx)(actx, tree, _force_generic=force_generic)
        htrav = actx.to_numpy(trav)
        assert_same_traversal(htrav, otrav)
    return htree, htrav


def test_c1_exact_recipe(actx, oracle):
    """configs[0]: 2D uniform, rng = default_rng(15); x, y = rng.random(10**5) each;
    sources = targets; max_particles_in_box = 30."""
    w, mpb = recipe("c1", 10**5)
    assert mpb == 30 and len(w["particles"]) == 2
    rng = np.random.default_rng(15)
    assert np.array_equal(w["particles"][0], rng.random(10**5))
    htree, htrav = build_and_compare(actx, oracle, w, mpb)
    check_tree(htree, w["particles"], max_particles_in_box=30)
    check_traversal(htree, htrav)


@pytest.mark.parametrize("workload,n", [("c2", 2 * 10**6), ("c3", 2 * 10**6),
                                        ("c3c", 2 * 10**6), ("c3c", 4 * 10**6)])
def test_point_recipes_against_oracle(actx, oracle, workload, n):
    """configs[1] and configs[2] (sphere surface, plain and clustered towards the
    poles) at oracle-sized samples: every Tree / FMMTraversalInfo array identical."""
    w, mpb = recipe(workload, n)
    htree, htrav = build_and_compare(actx, oracle, w, mpb, both_paths=(n <= 2 * 10**6))
    if workload == "c3c":
        # the clustered variant refines deeper than the plain sphere at the same size
        w3, _ = recipe("c3", n)
        o3 = oracle.build_tree(w3["particles"], max_particles_in_box=mpb)
        assert htree.nlevels >= o3.nlevels


def test_c4_recipe_against_oracle(actx, oracle):
    """configs[3] recipe (sources default_rng(15), targets default_rng(16), radii
    2**default_rng(12).uniform(-10, 0) * 2^-7, stick_out_factor 0.25, linf) at
    10^6 sources + 10^5 targets, both list paths."""
    w, mpb = recipe("c4", 10**6)
    assert np.array_equal(w["targets"][0], np.random.default_rng(16).random(10**5))
    htree, htrav = build_and_compare(actx, oracle, w, mpb)
    assert htree.targets_have_extent and htrav.from_sep_close_smaller_starts is not None
    check_traversal(htree, htrav)


def run_fmm(actx, tree, trav, weights):
    from boxtree_amd.constant_one import (ConstantOneExpansionWrangler,
                                          ConstantOneTreeIndependentDataForWrangler)
    from boxtree_amd.fmm import drive_fmm
    wrangler = ConstantOneExpansionWrangler(ConstantOneTreeIndependentDataForWrangler(), trav)
    return drive_fmm(actx, wrangler, (weights,))


def device_workload(workload):
    import torch
    from bench import make_workload
    return make_workload(torch, torch.device("cuda", 0), workload, None, 15)


def test_c4_full_size_completeness(actx):
    """configs[3] at full size: 10^8 sources, 10^7 targets with radii; every target
    receives 10^8."""
    import torch
    from boxtree_amd import FMMTraversalBuilder, TreeBuilder
    w = device_workload("c4")
    n = w["particles"][0].shape[0]
    nt = w["targets"][0].shape[0]
    assert n == 10**8 and nt == 10**7
    tree, _ = TreeBuilder(actx)(actx, w["particles"], targets=w["targets"],
                                max_particles_in_box=64, **w["kw"])
    trav, _ = FMMTraversalBuilder(actx)(actx, tree)
    assert trav.from_sep_close_smaller_starts is not None
    pot = run_fmm(actx, tree, trav, torch.ones(n, dtype=torch.float64, device="cuda"))
    assert pot.shape[0] == nt and int((pot != float(n)).sum()) == 0


def test_c3_clustered_full_size_completeness(actx):
    """The clustered variant of configs[2] at 10^8 points."""
    import torch
    from boxtree_amd import FMMTraversalBuilder, TreeBuilder
    w = device_workload("c3c")
    n = w["particles"][0].shape[0]
    assert n == 10**8
    tree, _ = TreeBuilder(actx)(actx, w["particles"], max_particles_in_box=64)
    trav, _ = FMMTraversalBuilder(actx)(actx, tree)
    pot = run_fmm(actx, tree, trav, torch.ones(n, dtype=torch.float64, device="cuda"))
    assert int((pot != float(n)).sum()) == 0
