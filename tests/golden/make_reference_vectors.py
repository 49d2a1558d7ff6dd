#!/usr/bin/env python
"""Generates tests/golden/reference_vectors.npz by RUNNING the reference.

Most of the reference cannot run in the build container: its kernels are OpenCL
inside Mako templates and pyopencl / arraycontext / pytools are absent (DESIGN.md
section 2).  A few pieces are plain Python + numpy, though, and those are executed
here, unmodified, from the files under /root/reference (which never travel to the
GPU box -- only the vectors written by this script do):

* ``boxtree.fmm.drive_fmm`` with ``boxtree.constant_one.ConstantOneExpansionWrangler``
  (fmm.py:342-532, constant_one.py:49-237) -- the reference's own consumer of a tree
  and its interaction lists.  It is run on the trees and traversals the CPU oracle
  builds for the cases of make_golden.py, with unit and with random weights, and
  the result of every stage is stored.  That the reference's consumer, reading
  the oracle's arrays by the reference's conventions, gives every target the total
  source weight is the acceptance test of test/test_fmm.py:141-391.
* ``AllReduceCommPattern`` (tools.py:756-855).
* ``get_box_ids_dfs_order`` and the root-rank loop of ``partition_work``
  (distributed/partition.py:39-121), run with a stand-in communicator object that
  records what ``Scatter`` hands to every rank.
* the per-stage loops of ``_PythonFMMCostModel`` (cost.py:1264-1360), with given
  per-level cost factors (the symbolic translation-cost model above them needs
  pymbolic, which is absent), on the oracle's trees and lists.
* ``boxtree.tree_of_boxes`` (make_tree_of_boxes_root, uniform and flagged refinement,
  coarsening, _sort_boxes_by_level; tree_of_boxes.py:123-465) -- plain numpy.  It is
  imported as a module; the ``TreeOfBoxes`` / ``box_flags_enum`` it imports from
  ``boxtree.tree`` (which needs pyopencl) are the field-for-field containers of
  boxtree_amd, ``pytools.single_valued`` is given its two-line meaning.  The pure-box
  trees it makes (numbered the reference's way: level by level, NOT in Morton
  order, every box flagged source and target, int32 levels, root parent -1) are
  stored as traversal INPUTS.
* ``RotationClassesBuilder.vec_gcd / compute_rotation_classes`` and
  ``TranslationClassesBuilder.ntranslation_classes_per_level /
  translation_class_to_normalized_vector`` (rotation_classes.py:102-162,
  translation_classes.py:302-322).

How they are loaded: ``boxtree.fmm`` and ``boxtree.constant_one`` are imported as
modules (a bare ``boxtree`` package object keeps ``boxtree/__init__.py``, which needs
pyopencl, from running; the one absent import they make, ``pytools.ProcessLogger``,
is a progress logger and is given a no-op here).  The other pieces live in modules
whose top-level imports need pyopencl/mako, so their function and class definitions
are compiled on their own from the parsed source file and executed with numpy as
the only global.  Nothing of the reference's text is written to the repository.

    python tests/golden/make_reference_vectors.py     # needs /root/reference
"""
import ast
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

OUT = os.path.join(HERE, "reference_vectors.npz")


# {{{ loading reference code

def definitions(relpath, names, inside=None):
    """Compiles the named top-level definitions of a reference file (or, with
    *inside*, the named methods of that class, as plain functions)."""
    path = os.path.join(REF, relpath)
    tree = ast.parse(open(path).read(), filename=path)
    body = tree.body
    if inside is not None:
        body = next(n for n in body if isinstance(n, ast.ClassDef) and n.name == inside).body
    picked = [n for n in body
              if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert sorted(n.name for n in picked) == sorted(names), (relpath, names)
    for n in picked:
        if isinstance(n, ast.FunctionDef):
            n.decorator_list = []          # @staticmethod / @log_process
            n.returns = None
            for a in n.args.args + n.args.kwonlyargs:
                a.annotation = None
    ns = {"np": np}
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return [ns[name] for name in names]


def import_fmm_and_constant_one():
    pkg = types.ModuleType("boxtree")
    pkg.__path__ = [os.path.join(REF, "boxtree")]
    sys.modules["boxtree"] = pkg

    class ProcessLogger:                   # pytools' progress logger: logging only
        def __init__(self, *args, **kwargs):
            pass

        def done(self, *args, **kwargs):
            pass

    pytools = types.ModuleType("pytools")
    pytools.ProcessLogger = ProcessLogger
    sys.modules["pytools"] = pytools
    sys.dont_write_bytecode = True
    import boxtree.constant_one as constant_one
    import boxtree.fmm as fmm
    return fmm, constant_one

# }}}


# {{{ drive_fmm + ConstantOneExpansionWrangler on the oracle's trees

def fmm_vectors(out):
    import make_golden as mg
    from oracle import oracle
    oracle.build_lib()
    fmm, constant_one = import_fmm_and_constant_one()

    class Recording(constant_one.ConstantOneExpansionWrangler):
        """Stores what each stage returns (the stage code itself is the reference's)."""
        def __init__(self, tree_indep, traversal):
            super().__init__(tree_indep, traversal)
            self.stages = {}

    def record(name):
        base = getattr(constant_one.ConstantOneExpansionWrangler, name)

        def method(self, *args, **kwargs):
            result = base(self, *args, **kwargs)
            n = sum(k.startswith(name) for k in self.stages)
            self.stages[f"{name}_{n}"] = np.array(result, copy=True)
            return result
        return method

    for name in ("form_multipoles", "coarsen_multipoles", "eval_direct", "multipole_to_local",
                 "eval_multipoles", "form_locals", "refine_locals", "eval_locals"):
        setattr(Recording, name, record(name))

    for name, case in mg.CASES.items():
        _inp, tree, trav = mg.build(oracle, case)
        trav.tree = tree
        rng = np.random.default_rng(1000 + case["seed"])
        for label, weights in (("ones", np.ones(tree.nsources)),
                               ("rand", rng.integers(1, 1000, tree.nsources).astype(np.float64))):
            wrangler = Recording(constant_one.ConstantOneTreeIndependentDataForWrangler(), trav)
            pot = fmm.drive_fmm(None, wrangler, [weights])
            # the reference's acceptance criterion (test/test_fmm.py:384-391)
            assert np.all(pot == weights.sum()), (name, label)
            out[f"fmm/{name}/{label}/weights"] = weights
            out[f"fmm/{name}/{label}/potentials"] = pot
            for k, v in wrangler.stages.items():
                out[f"fmm/{name}/{label}/{k}"] = v
        print(f"drive_fmm + ConstantOneExpansionWrangler on oracle case {name}: "
              f"potentials == total weight at all {tree.ntargets} targets")

# }}}


# {{{ AllReduceCommPattern

def comm_pattern_vectors(out):
    cls, = definitions("boxtree/tools.py", ["AllReduceCommPattern"])
    sizes = list(range(1, 41)) + [64, 100]
    rows = []            # size, rank, stage, sink, nsources, source0, source1, msg_lo, msg_hi
    for size in sizes:
        for rank in range(size):
            pat = cls(rank, size)
            stage = 0
            while not pat.done():
                sinks, sources = sorted(pat.sinks()), sorted(pat.sources())
                lo, hi = pat.messages()
                assert len(sinks) == 1 and len(sources) <= 2
                rows.append([size, rank, stage, sinks[0], len(sources),
                             sources[0] if sources else -1,
                             sources[1] if len(sources) > 1 else -1, lo, hi])
                pat.advance()
                stage += 1
    out["comm_pattern/rows"] = np.array(rows, dtype=np.int32)
    print(f"AllReduceCommPattern: {len(rows)} (size, rank, stage) rows")

# }}}


# {{{ depth-first order and partition_work

class _Comm:
    """Stands in for the MPI communicator: rank 0 of *size*; keeps what Scatter sends."""
    def __init__(self, size):
        self.size = size
        self.scattered = None

    def Get_rank(self):
        return 0

    def Get_size(self):
        return self.size

    def Scatter(self, sendbuf, recvbuf, root=0):
        self.scattered = np.array(sendbuf, copy=True)
        recvbuf[:] = sendbuf[0]


def partition_vectors(out):
    from types import SimpleNamespace

    from oracle import oracle
    oracle.build_lib()
    dfs, part = definitions("boxtree/distributed/partition.py",
                            ["get_box_ids_dfs_order", "partition_work"])
    part.__globals__["get_box_ids_dfs_order"] = dfs
    idx = 0
    for dims, n, mpb, seed in ((2, 300, 5, 1), (3, 2000, 10, 2), (3, 500, 3, 3), (2, 40, 1, 4)):
        rng = np.random.default_rng(seed)
        pts = [rng.standard_normal(n) for _ in range(dims)]
        tree = oracle.build_tree(pts, max_particles_in_box=mpb)
        rtree = SimpleNamespace(nboxes=tree.nboxes, dimensions=dims,
                                box_id_dtype=np.dtype(np.int32),
                                box_child_ids=tree.box_child_ids)
        order = dfs(rtree)
        trav = SimpleNamespace(tree=rtree)
        costs = {
            "random": rng.integers(0, 100, tree.nboxes).astype(np.float64),
            "ones": np.ones(tree.nboxes),
            "float": rng.random(tree.nboxes) * 3.7,
            "one_hot_last": np.eye(1, tree.nboxes, int(order[-1]))[0] * 9.0,
            "few": np.where(rng.random(tree.nboxes) < 0.01, 500.0, 0.0),
        }
        out[f"partition/{idx}/box_child_ids"] = tree.box_child_ids[:, :tree.nboxes]
        out[f"partition/{idx}/dims"] = np.array(dims)
        out[f"partition/{idx}/dfs_order"] = order
        for cname, cost in costs.items():
            out[f"partition/{idx}/cost/{cname}"] = cost
            for size in (1, 2, 3, 4, 7, 16):
                # segments left unassigned by the loop stay uninitialised upstream:
                # pre-fill what np.empty hands out so that they are recognisable
                comm = _Comm(size)
                mine = part(cost, trav, comm)
                seg = comm.scattered.astype(np.int64)
                assert np.array_equal(mine, order[seg[0, 0]:seg[0, 1]])
                # rows the loop wrote: all up to the last one with a plausible range
                written = 0
                start = 0
                for s in range(size):
                    if seg[s, 0] == start and start <= seg[s, 1] <= tree.nboxes and (
                            seg[s, 1] > seg[s, 0]):
                        written = s + 1
                        start = seg[s, 1]
                    else:
                        break
                out[f"partition/{idx}/segments/{cname}/{size}"] = seg[:written].astype(np.int32)
        idx += 1
    out["partition/ncases"] = np.array(idx)
    print(f"get_box_ids_dfs_order / partition_work: {idx} trees")

# }}}


# {{{ translation / rotation class arithmetic

def class_vectors(out):
    from types import SimpleNamespace
    nper, tovec = definitions("boxtree/translation_classes.py",
                              ["ntranslation_classes_per_level",
                               "translation_class_to_normalized_vector"],
                              inside="TranslationClassesBuilder")
    vec_gcd, compute = definitions("boxtree/rotation_classes.py",
                                   ["vec_gcd", "compute_rotation_classes"],
                                   inside="RotationClassesBuilder")
    tcb = SimpleNamespace(ntranslation_classes_per_level=nper)
    tcb.translation_class_to_normalized_vector = lambda n, d, c: tovec(tcb, n, d, c)
    rcb = SimpleNamespace(tcb=tcb, vec_gcd=vec_gcd)
    for nway in (1, 2, 3):
        for dims in (2, 3):
            ncls = nper(nway, dims)
            vecs = np.array([tovec(tcb, nway, dims, c) for c in range(ncls)])
            # the classes list 2 can produce: not adjacent, within 2n+1 boxes
            used = [c for c in range(ncls)
                    if np.max(np.abs(vecs[c])) > nway and np.max(np.abs(vecs[c])) <= 2 * nway + 1]
            rng = np.random.default_rng(nway * 10 + dims)
            some = sorted(rng.choice(used, size=max(1, len(used) // 3), replace=False).tolist())
            out[f"classes/{nway}_{dims}/vectors"] = vecs.astype(np.int32)
            for label, sel in (("all", used), ("some", some)):
                to_rot, angles = compute(rcb, nway, dims, sel)
                out[f"classes/{nway}_{dims}/{label}/used"] = np.array(sel, dtype=np.int32)
                out[f"classes/{nway}_{dims}/{label}/to_rot_class"] = to_rot
                out[f"classes/{nway}_{dims}/{label}/angles"] = np.array(angles, np.float64)
    print("translation / rotation class arithmetic: n-away 1..3, 2D and 3D")

# }}}


# {{{ per-stage loops of the Python cost model

COST_METHODS = ["process_form_multipoles", "get_ndirect_sources_per_target_box",
                "process_direct", "process_list2", "process_list3", "process_list4",
                "process_eval_locals", "process_coarsen_multipoles", "process_refine_locals"]


def cost_factors(case_seed, nlevels):
    """Integer-valued per-level factors (exactly summable in any order)."""
    rng = np.random.default_rng(5000 + case_seed)
    f = {k: rng.integers(1, 50, nlevels).astype(np.float64)
         for k in ("p2m_cost", "m2l_cost", "m2p_cost", "p2l_cost", "l2p_cost", "m2m_cost",
                   "l2l_cost")}
    f["c_p2p"] = np.float64(rng.integers(1, 9))
    return f


def cost_vectors(out):
    import make_golden as mg
    from oracle import oracle
    oracle.build_lib()
    fns = dict(zip(COST_METHODS, definitions("boxtree/cost.py", COST_METHODS,
                                             inside="_PythonFMMCostModel")))
    for name, case in mg.CASES.items():
        _inp, tree, trav = mg.build(oracle, case)
        trav.tree = tree
        f = cost_factors(case["seed"], tree.nlevels)
        pre = f"cost/{name}/"
        r = {}
        r["process_form_multipoles"] = fns["process_form_multipoles"](None, None, trav,
                                                                      f["p2m_cost"])
        nd = fns["get_ndirect_sources_per_target_box"](None, None, trav)
        r["get_ndirect_sources_per_target_box"] = nd
        r["process_direct"] = fns["process_direct"](None, None, trav, nd, f["c_p2p"])
        r["process_list2"] = fns["process_list2"](None, None, trav, f["m2l_cost"])
        r["process_list3"] = fns["process_list3"](None, None, trav, f["m2p_cost"])
        r["process_list4"] = fns["process_list4"](None, None, trav, f["p2l_cost"])
        r["process_eval_locals"] = fns["process_eval_locals"](None, None, trav, f["l2p_cost"])
        r["process_coarsen_multipoles"] = fns["process_coarsen_multipoles"](
            None, None, trav, f["m2m_cost"])
        r["process_refine_locals"] = fns["process_refine_locals"](None, None, trav,
                                                                  f["l2l_cost"])
        for k, v in r.items():
            out[pre + k] = np.asarray(v, dtype=np.float64)
    print(f"_PythonFMMCostModel stage loops on {len(mg.CASES)} oracle cases")

# }}}


# {{{ trees of boxes made by the reference

def tob_vectors(out):
    import dataclasses
    import operator

    from boxtree_amd.tree import TreeOfBoxes, box_flags_enum
    pkg = types.ModuleType("boxtree")
    pkg.__path__ = [os.path.join(REF, "boxtree")]
    tree_mod = types.ModuleType("boxtree.tree")
    tree_mod.TreeOfBoxes = TreeOfBoxes
    tree_mod.box_flags_enum = box_flags_enum
    pkg.tree = tree_mod

    def single_valued(iterable, equality_pred=operator.eq):
        it = iter(iterable)
        first = next(it)
        for other in it:
            assert equality_pred(other, first)
        return first

    pytools = types.ModuleType("pytools")
    pytools.single_valued = single_valued
    saved = {k: sys.modules.get(k) for k in ("boxtree", "boxtree.tree", "pytools",
                                             "boxtree.tree_of_boxes")}
    sys.modules.update({"boxtree": pkg, "boxtree.tree": tree_mod, "pytools": pytools})
    sys.modules.pop("boxtree.tree_of_boxes", None)
    try:
        import boxtree.tree_of_boxes as tobm
        _tob_cases(out, tobm, box_flags_enum)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _tob_cases(out, tobm, box_flags_enum):
    import dataclasses

    def store(name, tob):
        tob = tobm._sort_boxes_by_level(tob)
        for f in dataclasses.fields(tob):
            v = getattr(tob, f.name)
            if isinstance(v, np.ndarray):
                out[f"tob/{name}/{f.name}"] = v
        out[f"tob/{name}/root_extent"] = np.asarray(tob.root_extent)
        return tob

    names = []
    # test/test_tree_of_boxes.py:240-270: 2D, three uniform refinements
    radius = np.pi
    for dim, nlev, label in ((2, 3, "uniform_2d"), (3, 2, "uniform_3d"), (1, 4, "uniform_1d")):
        lower = np.zeros(dim) - radius / 2
        tob = tobm.make_tree_of_boxes_root((lower, lower + radius))
        for _ in range(nlev):
            tob = tobm.uniformly_refine_tree_of_boxes(tob)
        store(label, tob)
        names.append(label)
    # flagged refinement of random leaves, then some coarsening
    for dim, seed, label in ((2, 1, "adaptive_2d"), (3, 2, "adaptive_3d")):
        rng = np.random.default_rng(seed)
        tob = tobm.make_tree_of_boxes_root((np.zeros(dim), np.ones(dim)))
        tob = tobm.uniformly_refine_tree_of_boxes(tob)
        for _ in range(4 if dim == 2 else 3):
            leaf = tob.box_flags & box_flags_enum.IS_LEAF_BOX != 0
            flags = leaf & (rng.random(tob.nboxes) < 0.4)
            tob = tobm.refine_tree_of_boxes(tob, flags)
        store(label, tob)
        names.append(label)
        # coarsen: parents all of whose children are leaves, a random third of them
        child = tob.box_child_ids
        leaf = tob.box_flags & box_flags_enum.IS_LEAF_BOX != 0
        cand = [b for b in range(tob.nboxes)
                if not leaf[b] and all(leaf[c] for c in child[:, b] if c)]
        pick = [b for b in cand if rng.random() < 0.34]
        cflags = np.zeros(tob.nboxes, bool)
        for b in pick:
            cflags[[c for c in child[:, b] if c]] = True
        store(label + "_coarsened", tobm.coarsen_tree_of_boxes(tob, cflags))
        names.append(label + "_coarsened")
    out["tob/names"] = np.array(names)
    print(f"boxtree.tree_of_boxes: {len(names)} pure-box trees ({', '.join(names)})")

# }}}


def main():
    if not os.path.isdir(REF):
        raise SystemExit("needs the reference checkout at /root/reference")
    out = {}
    comm_pattern_vectors(out)
    partition_vectors(out)
    class_vectors(out)
    cost_vectors(out)
    tob_vectors(out)
    fmm_vectors(out)
    np.savez_compressed(OUT, **out)
    print(f"{len(out)} arrays -> {OUT} ({os.path.getsize(OUT) / 1e6:.2f} MB)")


if __name__ == "__main__":
    main()
