"""Translation / rotation classes on the device vs the oracle and the reference's
own check (test/test_traversal.py:327-403)."""

import numpy as np
import pytest

from test_oracle_classes import check_classes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def actx():
    from boxtree_amd import HIPArrayContext
    return HIPArrayContext(0)


@pytest.mark.parametrize("well_sep_is_n_away", [1, 2])
@pytest.mark.parametrize("dims,dtype", [(3, np.float64), (2, np.float64), (3, np.float32)])
def test_translation_and_rotation_classes(actx, oracle, well_sep_is_n_away, dims, dtype):
    from boxtree_amd import FMMTraversalBuilder, TreeBuilder
    from boxtree_amd.rotation_classes import RotationClassesBuilder
    from boxtree_amd.translation_classes import TranslationClassesBuilder
    rng = np.random.default_rng(15)
    p = [rng.normal(0.0, 1.0, 10**4).astype(dtype) for _ in range(dims)]
    tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(x) for x in p], max_particles_in_box=30)
    trav, _ = FMMTraversalBuilder(actx, well_sep_is_n_away=well_sep_is_n_away)(actx, tree)
    tc, _ = TranslationClassesBuilder(actx)(actx, trav, tree)
    rc, _ = RotationClassesBuilder(actx)(actx, trav, tree)
    otree = oracle.build_tree(p, max_particles_in_box=30)
    otrav = oracle.build_traversal(otree, well_sep_is_n_away=well_sep_is_n_away)
    otc = oracle.translation_classes(otree, otrav)
    orc = oracle.rotation_classes(otree, otrav)
    h, hr = actx.to_numpy(tc), actx.to_numpy(rc)
    for name in ("from_sep_siblings_translation_classes",
                 "from_sep_siblings_translation_class_to_distance_vector",
                 "from_sep_siblings_translation_classes_level_starts"):
        a, b = getattr(h, name), getattr(otc, name)
        assert a.dtype == b.dtype and np.array_equal(a, b), name
    assert np.array_equal(hr.from_sep_siblings_rotation_classes,
                          orc.from_sep_siblings_rotation_classes)
    assert np.array_equal(hr.from_sep_siblings_rotation_class_to_angle,
                          orc.from_sep_siblings_rotation_class_to_angle)
    assert tc.nfrom_sep_siblings_translation_classes == \
        otc.from_sep_siblings_translation_class_to_distance_vector.shape[1]
    if dtype == np.float64:
        check_classes(otree, otrav, h, hr)
    # classes shared by all levels
    tc0, _ = TranslationClassesBuilder(actx)(actx, trav, tree, is_translation_per_level=False)
    otc0 = oracle.translation_classes(otree, otrav, is_translation_per_level=False)
    h0 = actx.to_numpy(tc0)
    assert np.array_equal(h0.from_sep_siblings_translation_classes,
                          otc0.from_sep_siblings_translation_classes)
    assert np.array_equal(h0.from_sep_siblings_translation_classes_level_starts,
                          otc0.from_sep_siblings_translation_classes_level_starts)
