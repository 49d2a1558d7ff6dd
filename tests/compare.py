"""Field-by-field comparison of product output (moved to the host) with the
CPU oracle's output.  Integer/index arrays must be identical; coordinate arrays
must be identical too (np.array_equal: +0.0 == -0.0 is the only tolerance)."""

import numpy as np

TREE_SCALARS = ["sources_are_targets", "sources_have_extent", "targets_have_extent",
                "extent_norm", "nboxes", "nlevels", "aligned_nboxes", "nsources",
                "ntargets", "dimensions"]
TREE_ARRAYS = [
    "level_start_box_nrs", "user_source_ids", "sorted_target_ids",
    "box_source_starts", "box_source_counts_nonchild", "box_source_counts_cumul",
    "box_target_starts", "box_target_counts_nonchild", "box_target_counts_cumul",
    "box_parent_ids", "box_child_ids", "box_centers", "box_levels", "box_flags",
    "box_source_bounding_box_min", "box_source_bounding_box_max",
    "box_target_bounding_box_min", "box_target_bounding_box_max",
]


def _eq(name, a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.dtype == b.dtype, (name, a.dtype, b.dtype)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    if not np.array_equal(a, b):
        bad = np.argwhere(a != b)
        raise AssertionError(
            f"{name}: {len(bad)} mismatches, first at {bad[0]}: "
            f"{a[tuple(bad[0])]} != {b[tuple(bad[0])]}")


def assert_same_tree(t, o):
    for name in TREE_SCALARS:
        assert getattr(t, name) == getattr(o, name), (name, getattr(t, name), getattr(o, name))
    assert t.root_extent == o.root_extent
    assert np.dtype(t.coord_dtype) == np.dtype(o.coord_dtype)
    assert float(t.stick_out_factor) == float(o.stick_out_factor)
    for k in range(2):
        _eq(f"bounding_box[{k}]", t.bounding_box[k], o.bounding_box[k])
    for name in TREE_ARRAYS:
        _eq(name, getattr(t, name), getattr(o, name))
    for d in range(t.dimensions):
        _eq(f"sources[{d}]", t.sources[d], o.sources[d])
        _eq(f"targets[{d}]", t.targets[d], o.targets[d])
    for name in ("source_radii", "target_radii"):
        a, b = getattr(t, name), getattr(o, name)
        assert (a is None) == (b is None), name
        if a is not None:
            _eq(name, a, b)


TRAV_ARRAYS = [
    "source_boxes", "target_boxes", "source_parent_boxes", "target_or_target_parent_boxes",
    "level_start_source_box_nrs", "level_start_target_box_nrs",
    "level_start_source_parent_box_nrs", "level_start_target_or_target_parent_box_nrs",
    "same_level_non_well_sep_boxes_starts", "same_level_non_well_sep_boxes_lists",
    "neighbor_source_boxes_starts", "neighbor_source_boxes_lists",
    "from_sep_siblings_starts", "from_sep_siblings_lists",
    "from_sep_bigger_starts", "from_sep_bigger_lists",
]
TRAV_OPTIONAL = [
    "from_sep_close_smaller_starts", "from_sep_close_smaller_lists",
    "from_sep_close_bigger_starts", "from_sep_close_bigger_lists",
]


def assert_same_traversal(t, o):
    assert t.well_sep_is_n_away == o.well_sep_is_n_away
    for name in TRAV_ARRAYS:
        _eq(name, getattr(t, name), getattr(o, name))
    for name in TRAV_OPTIONAL:
        a, b = getattr(t, name), getattr(o, name)
        assert (a is None) == (b is None), name
        if a is not None:
            _eq(name, a, b)
    assert len(t.from_sep_smaller_by_level) == len(o.from_sep_smaller_by_level)
    for lev, (a, b) in enumerate(zip(t.from_sep_smaller_by_level,
                                     o.from_sep_smaller_by_level)):
        assert a.count == b.count, (lev, a.count, b.count)
        assert a.num_nonempty_lists == b.num_nonempty_lists, lev
        for name in ("starts", "lists", "nonempty_indices", "compressed_indices"):
            _eq(f"from_sep_smaller_by_level[{lev}].{name}", getattr(a, name), getattr(b, name))
        _eq(f"target_boxes_sep_smaller_by_source_level[{lev}]",
            t.target_boxes_sep_smaller_by_source_level[lev],
            o.target_boxes_sep_smaller_by_source_level[lev])
