"""Pins the CPU oracle against every invariant the reference's tests assert
(the reference holds no golden vectors; these property
checks are what its own suite uses).  CPU only."""

import numpy as np
import pytest

from invariants import check_traversal, check_tree, constant_one_potentials


def normal_particles(n, dims, dtype, seed=15):
    # boxtree/tools.py:114-119 make_normal_particle_array
    rng = np.random.default_rng(seed)
    return [rng.standard_normal(n, dtype=dtype) for _ in range(dims)]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("dims", [2, 3])
@pytest.mark.parametrize("n", [9, 4096, 10**5])
def test_bounding_box(oracle, dtype, dims, n):
    # test/test_tree.py:50-79
    p = normal_particles(n, dims, dtype)
    mn, mx = oracle.bounding_box(p)
    assert np.all(mn == [np.min(x) for x in p])
    assert np.all(mx == [np.max(x) for x in p])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("dims", [2, 3])
@pytest.mark.parametrize("n,mpb,kw", [
    (4, 30, {}),                       # test_single_box_particle_tree
    (50, 30, {}),                      # test_two_level_particle_tree
    (10**5, 30, {"skip_prune": True}),  # test_unpruned_particle_tree
    (10**5, 5, {}),                    # test_particle_tree_with_many_empty_leaves
    (10**5, 30, {}),                   # test_vanilla_particle_tree
    (10**4, 30, {"kind": "non-adaptive"}),  # test_non_adaptive_particle_tree
])
def test_particle_tree(oracle, dtype, dims, n, mpb, kw):
    p = normal_particles(n, dims, dtype)
    tree = oracle.build_tree(p, max_particles_in_box=mpb, **kw)
    check_tree(tree, p, max_particles_in_box=mpb)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("dims", [2, 3])
def test_explicit_refine_weights(oracle, dtype, dims):
    # test/test_tree.py:305-322
    n = 10**5
    p = normal_particles(n, dims, dtype)
    rw = np.random.default_rng(10).integers(1, 10, (n,), dtype=np.int32)
    tree = oracle.build_tree(p, refine_weights=rw, max_leaf_refine_weight=100)
    check_tree(tree, p, refine_weights=rw, max_leaf_refine_weight=100)


@pytest.mark.parametrize("dims", [2, 3])
def test_source_target_tree(oracle, dims):
    # test/test_tree.py:341-437
    s = normal_particles(2 * 10**5, dims, np.float64, seed=12)
    t = normal_particles(3 * 10**5, dims, np.float64, seed=19)
    tree = oracle.build_tree(s, targets=t, max_particles_in_box=10)
    check_tree(tree, s, targets=t, max_particles_in_box=10)


@pytest.mark.parametrize("dims", [2, 3])
@pytest.mark.parametrize("extent_norm", ["linf", "l2"])
def test_extent_tree(oracle, dims, extent_norm):
    # test/test_tree.py:445-629
    ns, nt = 100000, 200000
    s = normal_particles(ns, dims, np.float64, seed=12)
    t = normal_particles(nt, dims, np.float64, seed=19)
    rw = np.zeros(ns + nt, np.int32)
    rw[:ns] = 1
    rng = np.random.default_rng(13)
    sr = 2**rng.uniform(-10, 0, (ns,))
    tr = 2**rng.uniform(-10, 0, (nt,))
    tree = oracle.build_tree(s, targets=t, source_radii=sr, target_radii=tr,
                             extent_norm=extent_norm, refine_weights=rw,
                             max_leaf_refine_weight=20, stick_out_factor=0)
    check_tree(tree, s, targets=t, source_radii=sr, target_radii=tr,
               extent_norm=extent_norm)


def test_max_levels_exceeded(oracle):
    # test/test_tree.py:1103-1112
    p = [np.zeros(100), np.zeros(100)]
    p[0][:50] = 1
    with pytest.raises(oracle.MaxLevelsExceeded):
        oracle.build_tree(p, max_particles_in_box=10)


@pytest.mark.parametrize("dims,sat", [(2, True), (2, False), (3, True), (3, False)])
def test_tree_connectivity(oracle, dims, sat):
    # test/test_traversal.py:58-267
    s = normal_particles(10**5, dims, np.float64)
    t = None if sat else normal_particles(2 * 10**5, dims, np.float64)
    tree = oracle.build_tree(s, targets=t, max_particles_in_box=30)
    trav = oracle.build_traversal(tree)
    check_tree(tree, s, targets=t, max_particles_in_box=30)
    check_traversal(tree, trav)


@pytest.mark.parametrize("well_sep_is_n_away", [1, 2])
@pytest.mark.parametrize("dims,ns,nt,ext,extent_norm,crit", [
    (2, 10**5, None, "", "linf", "static_linf"),
    (2, 5 * 10**4, 4 * 10**4, "", "linf", "static_linf"),
    (2, 10**5, 4 * 10**4, "t", "linf", "static_linf"),
    (3, 10**5, None, "", "linf", "static_linf"),
    (3, 10**5, 4 * 10**4, "", "linf", "static_linf"),
    (3, 10**5, 4 * 10**4, "t", "linf", "static_linf"),
    (3, 10**5, 4 * 10**4, "t", "linf", "precise_linf"),
    (3, 10**5, 4 * 10**4, "t", "l2", "precise_linf"),
    (3, 10**5, 4 * 10**4, "t", "l2", "static_l2"),
])
def test_fmm_completeness(oracle, dims, ns, nt, ext, extent_norm, crit,
                          well_sep_is_n_away):
    # test/test_fmm.py:141-391 (sizes reduced to keep the CPU suite short)
    s = normal_particles(ns, dims, np.float64, seed=15)
    t = None if nt is None else normal_particles(nt, dims, np.float64, seed=16)
    rng = np.random.default_rng(12)
    tr = 2**rng.uniform(-10, 0, (nt,)) if "t" in ext else None
    tree = oracle.build_tree(s, targets=t, max_particles_in_box=30,
                             target_radii=tr, stick_out_factor=0.25,
                             extent_norm=extent_norm)
    trav = oracle.build_traversal(tree, well_sep_is_n_away=well_sep_is_n_away,
                                  from_sep_smaller_crit=crit)
    pot = constant_one_potentials(tree, trav)
    assert np.all(pot == ns)


def test_config_c1_exact_recipe(oracle):
    """BASELINE configs[0] (the reference's CPU-runnable case): 2D uniform,
    default_rng(15).random, 10^5 sources = targets, max_particles_in_box = 30 --
    tree invariants, traversal connectivity and interaction completeness."""
    from bench import WORKLOAD_MPB, make_workload_numpy
    w = make_workload_numpy("c1", 10**5, 15)
    mpb = WORKLOAD_MPB["c1"]
    tree = oracle.build_tree(w["particles"], max_particles_in_box=mpb)
    check_tree(tree, w["particles"], max_particles_in_box=mpb)
    trav = oracle.build_traversal(tree)
    check_traversal(tree, trav)
    pot = constant_one_potentials(tree, trav)
    assert np.all(pot == tree.nsources)
