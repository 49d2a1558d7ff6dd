"""The FMM call surface (boxtree/fmm.py:51-532) on the CPU: ``ExpansionWranglerInterface`` is
abstract where the reference's is, a wrangler that subclasses it -- here a numpy constant-one
wrangler written against the interface alone, on the oracle's tree and lists -- runs through
``drive_fmm``, and the stage table calls the methods in the reference's order."""

import inspect

import numpy as np
import pytest

from boxtree_amd.fmm import (FMM_STAGES, ExpansionWranglerInterface, TreeIndependentDataForWrangler,
                             drive_fmm)

# the abstract methods and the three hooks, with their parameter names (boxtree/fmm.py:136-338)
SURFACE = {
    "reorder_sources": ["source_array"], "reorder_potentials": ["potentials"],
    "multipole_expansions_view": ["mpole_exps", "level"], "local_expansions_view": ["local_exps", "level"],
    "form_multipoles": ["actx", "level_start_source_box_nrs", "source_boxes", "src_weight_vecs"],
    "coarsen_multipoles": ["actx", "level_start_source_parent_box_nrs", "source_parent_boxes", "mpoles"],
    "eval_direct": ["actx", "target_boxes", "neighbor_sources_starts", "neighbor_sources_lists",
                    "src_weight_vecs"],
    "multipole_to_local": ["actx", "level_start_target_or_target_parent_box_nrs",
                           "target_or_target_parent_boxes", "starts", "lists", "mpole_exps"],
    "eval_multipoles": ["actx", "target_boxes_by_source_level", "from_sep_smaller_by_level", "mpole_exps"],
    "form_locals": ["actx", "level_start_target_or_target_parent_box_nrs", "target_or_target_parent_boxes",
                    "starts", "lists", "src_weight_vecs"],
    "refine_locals": ["actx", "level_start_target_or_target_parent_box_nrs", "target_or_target_parent_boxes",
                      "local_exps"],
    "eval_locals": ["actx", "level_start_target_box_nrs", "target_boxes", "local_exps"],
    "finalize_potentials": ["actx", "potentials"],
    "distribute_source_weights": ["actx", "src_weight_vecs", "src_idx_all_ranks"],
    "gather_potential_results": ["actx", "potentials", "tgt_idx_all_ranks"],
    "communicate_mpoles": ["actx", "mpole_exps", "return_stats"],
}
HOOKS = {"distribute_source_weights", "gather_potential_results", "communicate_mpoles"}


def test_interface_surface():
    for name, params in SURFACE.items():
        fn = getattr(ExpansionWranglerInterface, name)
        assert list(inspect.signature(fn).parameters)[1:] == params, name
        assert bool(getattr(fn, "__isabstractmethod__", False)) == (name not in HOOKS), name
    with pytest.raises(TypeError, match="abstract"):
        ExpansionWranglerInterface(TreeIndependentDataForWrangler(), None)

    class Half(ExpansionWranglerInterface):
        def reorder_sources(self, source_array):
            return source_array

    with pytest.raises(TypeError, match="abstract"):
        Half(None, None)


class NumpyConstantOne(ExpansionWranglerInterface):
    """The constant-one kernel (every source contributes its weight to every target) against the
    interface: an expansion is one number per box."""

    calls = None

    def _rows(self, starts, lists, values):
        out = np.zeros(len(starts) - 1)
        if len(lists):
            np.add.at(out, np.repeat(np.arange(len(starts) - 1), np.diff(starts)), values[lists])
        return out

    def _box_weights(self, w):
        t = self.tree
        c = np.concatenate([[0.0], np.cumsum(w)])
        return c[t.box_source_starts + t.box_source_counts_nonchild] - c[t.box_source_starts]

    def _to_targets(self, boxes, per_box):
        t = self.tree
        pot = np.zeros(t.ntargets)
        for b, v in zip(boxes, per_box):
            s = t.box_target_starts[b]
            pot[s:s + t.box_target_counts_nonchild[b]] += v
        return pot

    def reorder_sources(self, source_array):
        return source_array[self.tree.user_source_ids]

    def reorder_potentials(self, potentials):
        return potentials[self.tree.sorted_target_ids]

    def multipole_expansions_view(self, mpole_exps, level):
        a, b = self.tree.level_start_box_nrs[level:level + 2]
        return a, mpole_exps[a:b]

    local_expansions_view = multipole_expansions_view

    def form_multipoles(self, actx, level_start_source_box_nrs, source_boxes, src_weight_vecs):
        self.calls.append("form_multipoles")
        m = np.zeros(self.tree.nboxes)
        m[source_boxes] = self._box_weights(src_weight_vecs[0])[source_boxes]
        return m

    def coarsen_multipoles(self, actx, level_start_source_parent_box_nrs, source_parent_boxes, mpoles):
        self.calls.append("coarsen_multipoles")
        t = self.tree
        for lev in range(t.nlevels - 1, 0, -1):
            a, b = level_start_source_parent_box_nrs[lev - 1:lev + 1]
            for ibox in source_parent_boxes[a:b]:
                ch = t.box_child_ids[:, ibox]
                mpoles[ibox] += mpoles[ch[ch != 0]].sum()
        return mpoles

    def eval_direct(self, actx, target_boxes, neighbor_sources_starts, neighbor_sources_lists, src_weight_vecs):
        self.calls.append("eval_direct")
        return self._to_targets(target_boxes, self._rows(neighbor_sources_starts, neighbor_sources_lists,
                                                         self._box_weights(src_weight_vecs[0])))

    def multipole_to_local(self, actx, level_start_target_or_target_parent_box_nrs,
                           target_or_target_parent_boxes, starts, lists, mpole_exps):
        self.calls.append("multipole_to_local")
        loc = np.zeros(self.tree.nboxes)
        loc[target_or_target_parent_boxes] = self._rows(starts, lists, mpole_exps)
        return loc

    def eval_multipoles(self, actx, target_boxes_by_source_level, from_sep_smaller_by_level, mpole_exps):
        self.calls.append("eval_multipoles")
        pot = np.zeros(self.tree.ntargets)
        for boxes, lst in zip(target_boxes_by_source_level, from_sep_smaller_by_level):
            pot += self._to_targets(boxes, self._rows(lst.starts, lst.lists, mpole_exps))
        return pot

    def form_locals(self, actx, level_start_target_or_target_parent_box_nrs, target_or_target_parent_boxes,
                    starts, lists, src_weight_vecs):
        self.calls.append("form_locals")
        loc = np.zeros(self.tree.nboxes)
        loc[target_or_target_parent_boxes] = self._rows(starts, lists, self._box_weights(src_weight_vecs[0]))
        return loc

    def refine_locals(self, actx, level_start_target_or_target_parent_box_nrs, target_or_target_parent_boxes,
                      local_exps):
        self.calls.append("refine_locals")
        t = self.tree
        for lev in range(1, t.nlevels):
            a, b = level_start_target_or_target_parent_box_nrs[lev:lev + 2]
            boxes = target_or_target_parent_boxes[a:b]
            local_exps[boxes] += local_exps[t.box_parent_ids[boxes]]
        return local_exps

    def eval_locals(self, actx, level_start_target_box_nrs, target_boxes, local_exps):
        self.calls.append("eval_locals")
        return self._to_targets(target_boxes, local_exps[target_boxes])

    def finalize_potentials(self, actx, potentials):
        self.calls.append("finalize_potentials")
        return potentials


@pytest.mark.parametrize("dims,extents", [(2, False), (3, False), (2, True), (3, True)])
def test_numpy_wrangler_through_drive_fmm(oracle, dims, extents):
    """Random integer weights in the caller's order: every target sees their sum -- and the
    methods run in the reference's order (fmm.py:380-532), the close lists only on trees with
    extents."""
    rng = np.random.default_rng(3 + dims)
    n = 3000
    pts = [rng.standard_normal(n) for _ in range(dims)]
    kw = {}
    if extents:
        kw = dict(targets=[rng.standard_normal(700) for _ in range(dims)],
                  target_radii=0.3 * 2.0 ** rng.uniform(-8, 0, 700), stick_out_factor=0.25)
    tree = oracle.build_tree(pts, max_particles_in_box=12, **kw)
    trav = oracle.build_traversal(tree)
    assert (trav.from_sep_close_smaller_starts is not None) == extents
    w = NumpyConstantOne(TreeIndependentDataForWrangler(), trav)
    assert w.tree is tree or w.tree.nboxes == tree.nboxes
    w.calls = []
    weights = rng.integers(1, 100, n).astype(np.float64)
    pot = drive_fmm(None, w, [weights])
    assert pot.shape == (tree.ntargets,) and np.all(pot == weights.sum())
    close = ["eval_direct"] if extents else []
    assert w.calls == (["form_multipoles", "coarsen_multipoles", "eval_direct", "multipole_to_local",
                        "eval_multipoles"] + close + ["form_locals"] + close
                       + ["refine_locals", "eval_locals", "finalize_potentials"])


def test_stage_table_is_the_reference_order():
    methods = [m for _, m, _, _, _ in FMM_STAGES]
    assert methods == ["form_multipoles", "coarsen_multipoles", "communicate_mpoles", "eval_direct",
                       "multipole_to_local", "eval_multipoles", "eval_direct", "form_locals", "eval_direct",
                       "refine_locals", "eval_locals"]
    for _, method, fields, reads, writes in FMM_STAGES:
        params = SURFACE[method]
        # traversal arrays + the one state input fill the method's parameters after `actx`
        assert len(fields) + 1 == len(params) - 1 - (1 if method == "communicate_mpoles" else 0), method
        assert reads in ("w", "m", "l") and writes in (None, "m", "l", "+l", "+p")


def test_particle_fixtures_on_the_host():
    """The reference's deterministic particle fixtures (boxtree/tools.py:122-276): sizes, the
    surfaces they lie on, and that an oracle tree of them passes the tree invariants."""
    from boxtree_amd.tools import surface_particle_coords, uniform_particle_coords
    x, y = surface_particle_coords(1000, 2, np.float64)
    assert x.shape == (1000,) and abs(x[0] - 1.5) < 1e-12 and abs(y[0]) < 1e-12
    x, y, z = surface_particle_coords(10000, 3, np.float64)
    assert x.shape == (10000,)
    rho = np.sqrt(x * x + y * y)
    assert np.allclose((rho - 15.0) ** 2 + z * z, 25.0)            # a torus of radii 15 and 5
    x, y = uniform_particle_coords(900, 2, np.float32)
    assert x.shape == (900,) and x.dtype == np.float32
    x, y, z = uniform_particle_coords(4096, 3, np.float64)
    assert x.shape == (15 ** 3,)                                    # int(4096 ** (1/3)) = 15
    # rotations keep distances: the lattice's nearest-neighbour spacing is 4 / (n - 1)
    d = np.sqrt((x[1] - x[0]) ** 2 + (y[1] - y[0]) ** 2 + (z[1] - z[0]) ** 2)
    assert abs(d - 4 / 14) < 1e-12
