"""Oracle translation / rotation classes pinned by the reference's check
(test/test_traversal.py:327-403)."""

import numpy as np
import pytest


def check_classes(tree, trav, tc, rc):
    dims = tree.dimensions
    centers = tree.box_centers.T
    classes = tc.from_sep_siblings_translation_classes
    vectors = tc.from_sep_siblings_translation_class_to_distance_vector
    assert classes.min(initial=0) >= 0
    for itgt, tgt in enumerate(trav.target_or_target_parent_boxes):
        s, e = trav.from_sep_siblings_starts[itgt:itgt + 2]
        seps = trav.from_sep_siblings_lists[s:e]
        if not len(seps):
            continue
        expected = centers[tgt] - centers[seps]
        assert np.allclose(vectors[:, classes[s:e]].T, expected, atol=1e-13, rtol=1e-13)
        if rc is not None:
            theta = np.arctan2(np.linalg.norm(expected[:, :dims - 1], axis=1), expected[:, dims - 1])
            got = rc.from_sep_siblings_rotation_class_to_angle[
                rc.from_sep_siblings_rotation_classes[s:e]]
            assert np.allclose(theta, got, atol=1e-13, rtol=1e-13)
    ls = tc.from_sep_siblings_translation_classes_level_starts
    assert np.all(np.diff(ls) >= 0) and ls[-1] == vectors.shape[1]


@pytest.mark.parametrize("well_sep_is_n_away", [1, 2])
@pytest.mark.parametrize("dims", [2, 3])
def test_oracle_translation_and_rotation_classes(oracle, well_sep_is_n_away, dims):
    rng = np.random.default_rng(15)
    p = [rng.normal(0.0, 1.0, 6000) for _ in range(dims)]
    tree = oracle.build_tree(p, max_particles_in_box=30)
    trav = oracle.build_traversal(tree, well_sep_is_n_away=well_sep_is_n_away)
    tc = oracle.translation_classes(tree, trav)
    rc = oracle.rotation_classes(tree, trav)
    check_classes(tree, trav, tc, rc)
    n = well_sep_is_n_away
    assert len(rc.from_sep_siblings_rotation_class_to_angle) <= 2 ** (dims - 1) * (2 * n + 1) ** dims
