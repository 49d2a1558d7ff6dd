"""Runs the reference's OWN test functions against the CPU oracle (test
infrastructure; used by tests/test_reference_suite.py, only where the reference
checkout exists -- it never travels to the GPU box).

The reference's tests build a tree / traversal through ``boxtree.TreeBuilder``,
``boxtree.traversal.FMMTraversalBuilder``, ... on an OpenCL array context, move the
result to the host with ``actx.to_numpy`` and then check it with plain numpy.  Here
the test functions are compiled, unmodified, from the files under
/root/reference/test (module-level imports of the absent packages dropped), and run
with

* an array context whose arrays simply are numpy arrays,
* a stand-in ``boxtree`` package whose builders call the oracle (same call
  signatures, host containers of boxtree_amd as return types) -- plus the
  reference's genuine ``boxtree.fmm`` and ``boxtree.constant_one`` modules, which
  are pure Python and are imported from the checkout,
* numpy particle generators in place of the pyopencl ones (test inputs, not logic).

Every assertion that then executes is the reference's.
"""

from __future__ import annotations

import ast
import contextlib
import logging
import os
import sys
import types
from types import SimpleNamespace

import numpy as np

REF = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF, "test"))


# {{{ array context stand-in

class _ActxNumpy:
    def zeros(self, shape, dtype):
        return np.zeros(shape, dtype)


class HostArrayContext:
    """Arrays are numpy arrays; every transfer is the identity."""
    np = _ActxNumpy()
    queue = SimpleNamespace(
        finish=lambda: None,
        device=SimpleNamespace(name="cpu oracle", platform=SimpleNamespace(name="none")))

    def to_numpy(self, x):
        return x

    def from_numpy(self, x):
        return x

    thaw = freeze = to_numpy

# }}}


# {{{ containers

def obj_array_1d(arrays):
    out = np.empty(len(arrays), dtype=object)
    for i, a in enumerate(arrays):
        out[i] = a
    return out


def host_tree(t):
    """The oracle's namespace as the product's (numpy-backed) Tree container, which
    carries the reference's helper methods (get_box_extent, ...)."""
    import dataclasses

    from boxtree_amd.tree import Tree
    kw = {}
    for f in dataclasses.fields(Tree):
        kw[f.name] = getattr(t, f.name)
    kw["sources"] = obj_array_1d(list(t.sources))
    kw["targets"] = kw["sources"] if t.sources_are_targets else obj_array_1d(list(t.targets))
    return Tree(**kw)


class HostTraversal(SimpleNamespace):
    def merge_close_lists(self, actx, debug=False):
        """FMMTraversalInfo.merge_close_lists (traversal.py:1650-1693): list 1, then
        close-smaller, then close-bigger, per target box."""
        parts = [(self.neighbor_source_boxes_starts, self.neighbor_source_boxes_lists),
                 (self.from_sep_close_smaller_starts, self.from_sep_close_smaller_lists),
                 (self.from_sep_close_bigger_starts, self.from_sep_close_bigger_lists)]
        ntb = len(self.target_boxes)
        starts = np.zeros(ntb + 1, np.int32)
        lists = []
        for i in range(ntb):
            for st, li in parts:
                lists.extend(li[st[i]:st[i + 1]])
            starts[i + 1] = len(lists)
        new = HostTraversal(**self.__dict__)
        new.neighbor_source_boxes_starts = starts
        new.neighbor_source_boxes_lists = np.array(lists, dtype=np.int32)
        for name in ("from_sep_close_smaller", "from_sep_close_bigger"):
            setattr(new, name + "_starts", None)
            setattr(new, name + "_lists", None)
        return new

# }}}


# {{{ the stand-in ``boxtree`` package

def _oracle_tree_of(tree):
    return tree._oracle


def make_boxtree_modules(oracle):
    from boxtree_amd.tree import box_flags_enum

    class TreeBuilder:
        def __init__(self, array_context):
            pass

        def __call__(self, actx, particles, kind="adaptive", max_particles_in_box=None,
                     allocator=None, debug=False, targets=None, source_radii=None,
                     target_radii=None, stick_out_factor=None, refine_weights=None,
                     max_leaf_refine_weight=None, wait_for=None, extent_norm=None, bbox=None,
                     **kwargs):
            kwargs.pop("nboxes_guess", None)          # an allocation hint upstream
            t = oracle.build_tree(
                list(particles), kind=kind, max_particles_in_box=max_particles_in_box,
                targets=None if targets is None else list(targets),
                source_radii=source_radii, target_radii=target_radii,
                stick_out_factor=stick_out_factor, refine_weights=refine_weights,
                max_leaf_refine_weight=max_leaf_refine_weight, extent_norm=extent_norm,
                bbox=bbox, **kwargs)
            tree = host_tree(t)
            object.__setattr__(tree, "_oracle", t)
            return tree, None

    class FMMTraversalBuilder:
        def __init__(self, array_context, *, well_sep_is_n_away=1, from_sep_smaller_crit=None):
            self.kw = dict(well_sep_is_n_away=well_sep_is_n_away,
                           from_sep_smaller_crit=from_sep_smaller_crit)

        def __call__(self, actx, tree, wait_for=None, debug=False,
                     _from_sep_smaller_min_nsources_cumul=None, source_boxes_mask=None,
                     source_parent_boxes_mask=None):
            r = oracle.build_traversal(
                _oracle_tree_of(tree),
                _from_sep_smaller_min_nsources_cumul=_from_sep_smaller_min_nsources_cumul,
                source_boxes_mask=source_boxes_mask,
                source_parent_boxes_mask=source_parent_boxes_mask, **self.kw)
            trav = HostTraversal(**r.__dict__)
            trav.tree = tree
            trav.well_sep_is_n_away = self.kw["well_sep_is_n_away"]
            return trav, None

    class ParticleListFilter:
        def __init__(self, array_context):
            pass

        def filter_target_lists_in_user_order(self, actx, tree, flags):
            return oracle.filter_target_lists_in_user_order(_oracle_tree_of(tree), flags)

        def filter_target_lists_in_tree_order(self, actx, tree, flags):
            return oracle.filter_target_lists_in_tree_order(_oracle_tree_of(tree), flags)

    def _balls(ball_centers, ball_radii):
        return [np.asarray(c) for c in ball_centers], np.asarray(ball_radii)

    class AreaQueryBuilder:
        def __init__(self, array_context):
            pass

        def __call__(self, actx, tree, ball_centers, ball_radii, peer_lists=None,
                     wait_for=None):
            c, r = _balls(ball_centers, ball_radii)
            return oracle.area_query(_oracle_tree_of(tree), c, r), None

    class LeavesToBallsLookupBuilder(AreaQueryBuilder):
        def __call__(self, actx, tree, ball_centers, ball_radii, wait_for=None):
            c, r = _balls(ball_centers, ball_radii)
            return oracle.leaves_to_balls(_oracle_tree_of(tree), c, r), None

    class SpaceInvaderQueryBuilder(AreaQueryBuilder):
        def __call__(self, actx, tree, ball_centers, ball_radii, peer_lists=None,
                     wait_for=None):
            c, r = _balls(ball_centers, ball_radii)
            return oracle.space_invader_query(_oracle_tree_of(tree), c, r), None

    class PeerListFinder(AreaQueryBuilder):
        def __call__(self, actx, tree, wait_for=None):
            return oracle.peer_lists(_oracle_tree_of(tree)), None

    class TranslationClassesBuilder:
        def __init__(self, array_context):
            pass

        def __call__(self, actx, trav, tree, wait_for=None, is_translation_per_level=True):
            return oracle.translation_classes(
                _oracle_tree_of(tree), trav,
                is_translation_per_level=is_translation_per_level), None

    class RotationClassesBuilder:
        def __init__(self, array_context):
            pass

        def __call__(self, actx, trav, tree, wait_for=None):
            return oracle.rotation_classes(_oracle_tree_of(tree), trav), None

    def make_normal_particle_array(actx, nparticles, dims, dtype, seed=15):
        rng = np.random.default_rng(seed)
        return obj_array_1d([rng.standard_normal(nparticles).astype(dtype)
                             for _ in range(dims)])

    def make_uniform_particle_array(actx, nparticles, dims, dtype, seed=15):
        rng = np.random.default_rng(seed)
        return obj_array_1d([rng.random(nparticles).astype(dtype) for _ in range(dims)])

    def make_surface_particle_array(actx, nparticles, dims, dtype, seed=15):
        # upstream: a torus/ellipse lattice generated with loopy; any surface will do
        # as a test input
        rng = np.random.default_rng(seed)
        v = rng.standard_normal((dims, nparticles))
        v /= np.sqrt((v * v).sum(axis=0))
        v *= np.array([1.0, 0.6, 0.8][:dims])[:, None]
        return obj_array_1d([np.ascontiguousarray(v[i]).astype(dtype) for i in range(dims)])

    def particle_array_to_host(parray):
        return np.array(list(parray)).T

    pkg = types.ModuleType("boxtree")
    pkg.__path__ = [os.path.join(REF, "boxtree")]       # fmm / constant_one: genuine
    pkg.TreeBuilder = TreeBuilder
    pkg.box_flags_enum = box_flags_enum
    mods = {"boxtree": pkg}

    def sub(name, **attrs):
        m = types.ModuleType("boxtree." + name)
        m.__dict__.update(attrs)
        setattr(pkg, name, m)
        mods["boxtree." + name] = m

    sub("traversal", FMMTraversalBuilder=FMMTraversalBuilder)
    sub("tree", ParticleListFilter=ParticleListFilter, box_flags_enum=box_flags_enum,
        link_point_sources=lambda actx, tree, pss, ps, debug=False: oracle.link_point_sources(
            _oracle_tree_of(tree), pss, ps))
    sub("tree_build", MaxLevelsExceeded=oracle.MaxLevelsExceeded, TreeBuilder=TreeBuilder)
    sub("area_query", AreaQueryBuilder=AreaQueryBuilder,
        LeavesToBallsLookupBuilder=LeavesToBallsLookupBuilder,
        SpaceInvaderQueryBuilder=SpaceInvaderQueryBuilder, PeerListFinder=PeerListFinder)
    sub("translation_classes", TranslationClassesBuilder=TranslationClassesBuilder)
    sub("rotation_classes", RotationClassesBuilder=RotationClassesBuilder)
    sub("tools", make_normal_particle_array=make_normal_particle_array,
        make_uniform_particle_array=make_uniform_particle_array,
        make_surface_particle_array=make_surface_particle_array,
        particle_array_to_host=particle_array_to_host, AXIS_NAMES=("x", "y", "z", "w"))

    class ProcessLogger:                      # pytools' progress logger
        def __init__(self, *a, **k):
            pass

        def done(self, *a, **k):
            pass

    pytools = types.ModuleType("pytools")
    pytools.ProcessLogger = ProcessLogger
    pytools.obj_array = SimpleNamespace(new_1d=obj_array_1d)
    mods["pytools"] = pytools
    return mods


@contextlib.contextmanager
def installed(mods):
    saved = {k: sys.modules.get(k) for k in list(mods) + ["boxtree.fmm", "boxtree.constant_one"]}
    dont = sys.dont_write_bytecode
    sys.dont_write_bytecode = True
    sys.modules.update(mods)
    try:
        yield
    finally:
        sys.dont_write_bytecode = dont
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v

# }}}


# {{{ compiling the reference's test files

_DROP_IMPORTS = ("arraycontext", "pyopencl", "boxtree.array_context", "pytools")


def load_test_module(relpath, oracle):
    """-> (namespace of the test file's definitions, module dict to install)."""
    import pytest
    mods = make_boxtree_modules(oracle)
    path = os.path.join(REF, relpath)
    tree = ast.parse(open(path).read(), filename=path)
    body = []
    for node in tree.body:
        if isinstance(node, ast.ImportFrom) and node.module and (
                node.module.startswith(_DROP_IMPORTS)):
            continue
        if isinstance(node, ast.Import) and any(
                a.name.startswith(_DROP_IMPORTS) for a in node.names):
            continue
        if isinstance(node, ast.Assign) and any(
                isinstance(t, ast.Name) and t.id == "pytest_generate_tests"
                for t in node.targets):
            continue
        if isinstance(node, ast.If):           # the ``if __name__ == "__main__"`` runner
            continue
        body.append(node)
    ns = {"__name__": "reference_" + os.path.basename(relpath)[:-3],
          "obj_array": mods["pytools"].obj_array, "pytest": pytest,
          "logging": logging, "np": np}
    with installed(mods):
        exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
        for k in ("boxtree.fmm", "boxtree.constant_one"):     # genuine, imported above
            if k in sys.modules:
                mods[k] = sys.modules[k]
    return ns, mods


def call(ns, mods, name, **kwargs):
    """Calls reference test function *name* (undecorated) with an ``actx_factory``."""
    fn = ns[name]
    fn = getattr(fn, "__wrapped__", fn)
    with installed(mods):
        if "actx_factory" in fn.__code__.co_varnames[:fn.__code__.co_argcount]:
            kwargs["actx_factory"] = HostArrayContext
        return fn(**kwargs)

# }}}
