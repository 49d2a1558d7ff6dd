"""The reference's own test functions, compiled from /root/reference/test at run
time and executed against the CPU oracle (see tests/reference_suite.py).  Skipped
where the reference checkout is absent (the GPU box)."""

import os

import numpy as np
import pytest

import reference_suite as rs

pytestmark = pytest.mark.skipif(not rs.available(), reason="needs /root/reference")

_LOADED = {}


def _module(relpath, oracle):
    if relpath not in _LOADED:
        _LOADED[relpath] = rs.load_test_module(relpath, oracle)
    return _LOADED[relpath]


PARTICLE_TREE_TESTS = [
    "test_single_box_particle_tree", "test_two_level_particle_tree",
    "test_unpruned_particle_tree", "test_particle_tree_with_reallocations",
    "test_particle_tree_with_many_empty_leaves", "test_vanilla_particle_tree",
    "test_explicit_refine_weights_particle_tree", "test_non_adaptive_particle_tree",
]


@pytest.mark.parametrize("dims", [2, 3])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("name", PARTICLE_TREE_TESTS)
def test_reference_particle_tree_tests(oracle, name, dtype, dims):
    """test/test_tree.py:229-334 (each through run_build_test, :86-226)."""
    ns, mods = _module("test/test_tree.py", oracle)
    rs.call(ns, mods, name, dtype=dtype, dims=dims)


@pytest.mark.parametrize("dims", [2, 3])
def test_reference_source_target_tree(oracle, dims):
    """test/test_tree.py:341-444."""
    ns, mods = _module("test/test_tree.py", oracle)
    rs.call(ns, mods, "test_source_target_tree", dims=dims)


@pytest.mark.parametrize("extent_norm", ["linf", "l2"])
@pytest.mark.parametrize("dims", [2, 3])
def test_reference_extent_tree(oracle, dims, extent_norm):
    """test/test_tree.py:451-665 (includes link_point_sources)."""
    ns, mods = _module("test/test_tree.py", oracle)
    rs.call(ns, mods, "test_extent_tree", dims=dims, extent_norm=extent_norm)


@pytest.mark.parametrize("dims", [2, 3])
def test_reference_leaves_to_balls_query(oracle, dims):
    """test/test_tree.py:672-724."""
    ns, mods = _module("test/test_tree.py", oracle)
    rs.call(ns, mods, "test_leaves_to_balls_query", dims=dims)


@pytest.mark.parametrize("dims", [2, 3])
@pytest.mark.parametrize("name", ["test_area_query", "test_area_query_balls_outside_bbox"])
def test_reference_area_query(oracle, name, dims):
    """test/test_tree.py:730-835."""
    ns, mods = _module("test/test_tree.py", oracle)
    rs.call(ns, mods, name, dims=dims)


@pytest.mark.parametrize("dims", [2, 3])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_reference_space_invader_query(oracle, dims, dtype):
    """test/test_tree.py:985-1041."""
    ns, mods = _module("test/test_tree.py", oracle)
    rs.call(ns, mods, "test_space_invader_query", dims=dims, dtype=dtype)


@pytest.mark.parametrize("dims", [2, 3])
def test_reference_same_tree_with_zero_weight_particles(oracle, dims):
    """test/test_tree.py:1050-1097."""
    ns, mods = _module("test/test_tree.py", oracle)
    rs.call(ns, mods, "test_same_tree_with_zero_weight_particles", dims=dims)


def test_reference_max_levels_error(oracle):
    """test/test_tree.py:1103-1114."""
    ns, mods = _module("test/test_tree.py", oracle)
    rs.call(ns, mods, "test_max_levels_error")


@pytest.mark.parametrize("sources_are_targets", [True, False])
@pytest.mark.parametrize("dims", [2, 3])
def test_reference_tree_connectivity(oracle, dims, sources_are_targets):
    """test/test_traversal.py:58-272."""
    ns, mods = _module("test/test_traversal.py", oracle)
    rs.call(ns, mods, "test_tree_connectivity", dims=dims,
            sources_are_targets=sources_are_targets)


@pytest.mark.parametrize("lookbehind", [0, 1])
@pytest.mark.parametrize("skip_prune", [True, False])
@pytest.mark.parametrize("dims", [2, 3])
def test_reference_level_restriction(oracle, dims, skip_prune, lookbehind):
    """test/test_tree.py:900-974."""
    ns, mods = _module("test/test_tree.py", oracle)
    rs.call(ns, mods, "test_level_restriction", dims=dims, skip_prune=skip_prune,
            lookbehind=lookbehind)


@pytest.mark.parametrize("well_sep_is_n_away", [1, 2])
def test_reference_translation_and_rotation_classes(oracle, well_sep_is_n_away):
    """test/test_traversal.py:327-410."""
    ns, mods = _module("test/test_traversal.py", oracle)
    rs.call(ns, mods, "test_from_sep_siblings_translation_and_rotation_classes",
            well_sep_is_n_away=well_sep_is_n_away)


@pytest.mark.parametrize("well_sep_is_n_away", [1, 2])
@pytest.mark.parametrize("icase", range(14))
def test_reference_fmm_completeness(oracle, icase, well_sep_is_n_away):
    """test/test_fmm.py:141-391: the reference's drive_fmm + constant-one wranglers
    (with its filtered-target variants) on the oracle's trees and lists."""
    ns, mods = _module("test/test_fmm.py", oracle)
    fn = ns["test_fmm_completeness"]
    marks = [m for m in fn.pytestmark if m.name == "parametrize"]
    cases = next(m for m in marks if isinstance(m.args[0], tuple))
    names, values = cases.args
    assert len(values) == 14
    kwargs = dict(zip(names, values[icase]))
    if kwargs["nsources_req"] > 10**5 and os.environ.get("BOXTREE_REFERENCE_SUITE") != "full":
        # the reference's pure-Python wrangler needs 10-30 s on the 5*10^5-source
        # cases; the full run is logged in tests/golden/reference_suite_full.txt
        pytest.skip("set BOXTREE_REFERENCE_SUITE=full for the 5*10^5-source cases")
    rs.call(ns, mods, "test_fmm_completeness", well_sep_is_n_away=well_sep_is_n_away,
            **kwargs)
