"""FMM cost model on the device vs the numpy restatement of the reference's
_PythonFMMCostModel (boxtree/cost.py:1264-1440; the reference's own test compares
its two implementations the same way, test/test_cost_model.py)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def actx():
    from boxtree_amd import HIPArrayContext
    return HIPArrayContext(0)


@pytest.mark.parametrize("dims,ntargets,extent,taylor", [
    (2, None, False, False), (3, None, False, False), (3, 5000, True, False),
    (3, None, False, True)])
def test_cost_model_matches_python_statement(actx, oracle, dims, ntargets, extent, taylor):
    from boxtree_amd import FMMTraversalBuilder, TreeBuilder
    from boxtree_amd.cost import (FMMCostModel, make_pde_aware_translation_cost_model,
                                  make_taylor_translation_cost_model)
    rng = np.random.default_rng(9)
    n = 20000
    src = [rng.standard_normal(n) for _ in range(dims)]
    kw = dict(max_particles_in_box=30)
    okw = dict(kw)
    if ntargets:
        tgt = [rng.standard_normal(ntargets) for _ in range(dims)]
        kw["targets"] = [actx.from_numpy(t) for t in tgt]
        okw["targets"] = tgt
    if extent:
        radii = 2.0 ** rng.uniform(-10, 0, ntargets)
        kw.update(target_radii=actx.from_numpy(radii), stick_out_factor=0.25)
        okw.update(target_radii=radii, stick_out_factor=0.25)
    tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(s) for s in src], **kw)
    trav, _ = FMMTraversalBuilder(actx)(actx, tree)
    otree = oracle.build_tree(src, **okw)
    otrav = oracle.build_traversal(otree)
    nlevels = int(tree.nlevels)
    level_to_order = np.array([3 + (lev % 4) for lev in range(nlevels)])
    params = dict(c_l2l=1.5, c_l2p=0.5, c_m2l=2.0, c_m2m=1.25, c_m2p=0.75, c_p2l=3.0,
                  c_p2m=1.0, c_p2p=0.125)
    factory = make_taylor_translation_cost_model if taylor else make_pde_aware_translation_cost_model
    model = FMMCostModel(factory)
    per_box = model.cost_per_box(actx, trav, level_to_order, dict(params))
    per_stage = model.cost_per_stage(actx, trav, level_to_order, dict(params))
    want_box, want_stage = oracle.cost_model(otree, otrav, level_to_order, params, taylor=taylor)
    got = per_box.cpu().numpy()
    assert got.shape == want_box.shape
    assert np.allclose(got, want_box, rtol=1e-13, atol=0)
    assert set(per_stage) == set(want_stage)
    for stage, v in want_stage.items():
        assert np.isclose(per_stage[stage], v, rtol=1e-12), stage
    # the per-box costs add up to the per-stage costs minus the two tree sweeps
    total = sum(v for k, v in want_stage.items()
                if k not in ("coarsen_multipoles", "refine_locals"))
    assert np.isclose(got.sum(), total, rtol=1e-12)


def test_calibration_recovers_factors(actx):
    """test_cost_model.py::test_estimate_calibration_params in spirit: timings that
    are exact multiples of the unit-parameter model give those multiples back."""
    from boxtree_amd import FMMTraversalBuilder, TreeBuilder
    from boxtree_amd.cost import FMMCostModel
    model = FMMCostModel()
    truth = dict(c_l2l=2e-9, c_l2p=3e-9, c_m2l=5e-9, c_m2m=7e-9, c_m2p=1.1e-8, c_p2l=1.3e-8,
                 c_p2m=1.7e-8, c_p2p=1.9e-8)
    stage_param = model._FMM_STAGE_TO_CALIBRATION_PARAMETER
    model_results, timing_results = [], []
    for seed, n in ((1, 5000), (2, 9000), (3, 14000)):
        rng = np.random.default_rng(seed)
        src = [actx.from_numpy(rng.standard_normal(n)) for _ in range(3)]
        tree, _ = TreeBuilder(actx)(actx, src, max_particles_in_box=20)
        trav, _ = FMMTraversalBuilder(actx)(actx, tree)
        order = np.full(int(tree.nlevels), 4)
        res = model.cost_per_stage(actx, trav, order, model.get_unit_calibration_params())
        model_results.append(res)
        timing_results.append({stage: {"wall_elapsed": res[stage] * truth[stage_param[stage]]}
                               for stage in res})
    est = model.estimate_calibration_params(model_results, timing_results)
    for name, v in truth.items():
        assert np.isclose(est[name], v, rtol=1e-12), name
