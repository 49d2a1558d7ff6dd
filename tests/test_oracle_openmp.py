"""The OpenMP build of the oracle (bench.py's all-core CPU baseline) against the
sequential build: same source, chunked loops, identical arrays.  CPU only."""

import numpy as np
import pytest

from compare import assert_same_traversal, assert_same_tree


@pytest.fixture()
def both(oracle):
    yield oracle
    oracle.set_variant("seq")


def build(oracle, variant, threads, p, targets=None, trav_kw=None, **kw):
    oracle.set_variant(variant, threads)
    tree = oracle.build_tree(p, targets=targets, **kw)
    trav = oracle.build_traversal(tree, **(trav_kw or {}))
    return tree, trav


@pytest.mark.parametrize("dims,n,threads", [(2, 30000, 3), (3, 60000, 7), (3, 60000, 8)])
def test_openmp_point_tree(both, dims, n, threads):
    rng = np.random.default_rng(dims * 100 + threads)
    p = [rng.standard_normal(n) for _ in range(dims)]
    t0, v0 = build(both, "seq", None, p, max_particles_in_box=30)
    t1, v1 = build(both, "omp", threads, p, max_particles_in_box=30)
    assert_same_tree(t1, t0)
    assert_same_traversal(v1, v0)


def test_openmp_extent_tree(both):
    rng = np.random.default_rng(4)
    src = [rng.random(40000) for _ in range(3)]
    tgt = [rng.random(9000) for _ in range(3)]
    radii = 2.0 ** rng.uniform(-10, 0, 9000) * 2.0 ** -4
    kw = dict(targets=tgt, target_radii=radii, stick_out_factor=0.25, max_particles_in_box=20)
    t0, v0 = build(both, "seq", None, src, **kw)
    t1, v1 = build(both, "omp", 5, src, **kw)
    assert_same_tree(t1, t0)
    assert_same_traversal(v1, v0)


def test_openmp_weights_and_level_restriction(both):
    rng = np.random.default_rng(8)
    p = [rng.standard_normal(30000) for _ in range(2)]
    rw = rng.integers(0, 6, 30000, dtype=np.int32)
    kw = dict(refine_weights=rw, max_leaf_refine_weight=40, kind="adaptive-level-restricted")
    t0, v0 = build(both, "seq", None, p, **kw)
    t1, v1 = build(both, "omp", 6, p, **kw)
    assert_same_tree(t1, t0)
    assert_same_traversal(v1, v0)
