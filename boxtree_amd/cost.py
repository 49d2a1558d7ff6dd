"""FMM cost model: operation counts and calibrated cost estimates per box and per
stage for a traversal (boxtree/cost.py:87-1444).

The per-box processors run on the device: interaction-list sums go through the
library's CSR kernel (``bt_fmm_csr_sum``), the remaining per-box products are
elementwise tensor arithmetic on arrays of ``nboxes`` entries.
"""

from __future__ import annotations

import numpy as np

from boxtree_amd import _lib
from boxtree_amd.array_context import ptr

__all__ = [
    "AbstractFMMCostModel", "FMMCostModel", "FMMTranslationCostModel",
    "make_pde_aware_translation_cost_model", "make_taylor_translation_cost_model",
]


# {{{ tiny symbolic layer (the reference uses pymbolic variables, cost.py:101-150)

class _Expr:
    """A polynomial-free arithmetic expression tree over named variables."""

    def __init__(self, op, args):
        self.op, self.args = op, args

    def evaluate(self, context):
        if self.op == "var":
            return context[self.args[0]]
        vals = [a.evaluate(context) if isinstance(a, _Expr) else a for a in self.args]
        if self.op == "+":
            return vals[0] + vals[1]
        if self.op == "*":
            return vals[0] * vals[1]
        if self.op == "**":
            return vals[0] ** vals[1]
        raise ValueError(self.op)

    def __add__(self, other):
        return _Expr("+", (self, other))

    __radd__ = __add__

    def __mul__(self, other):
        return _Expr("*", (self, other))

    __rmul__ = __mul__

    def __pow__(self, other):
        return _Expr("**", (self, other))


def var(name):
    return _Expr("var", (name,))


def evaluate(expr, context):
    return expr.evaluate(context) if isinstance(expr, _Expr) else expr

# }}}


# {{{ translation cost model (cost.py:87-183)

# Every translation is "calibration constant x size term".  The size term of a
# particle <-> expansion operator is the coefficient count of its level; that of an
# expansion -> expansion operator couples the counts of both levels (product, or the
# rotate-translate-rotate sum when point-and-shoot is used).
_PARTICLE_OPERATORS = ("p2l", "l2p", "p2m", "m2p")          # (level)
_EXPANSION_OPERATORS = ("m2m", "l2l", "m2l")                # (source level, target level)


def _level_operator(name):
    def cost(self, level):
        return var("c_" + name) * self.ncoeffs_fmm_by_level[level]
    cost.__name__ = name
    return cost


def _pair_operator(name):
    def cost(self, src_level, tgt_level):
        counts = self.ncoeffs_fmm_by_level
        return var("c_" + name) * self.e2e_cost(counts[src_level], counts[tgt_level])
    cost.__name__ = name
    return cost


class FMMTranslationCostModel:
    """Modeled costs of the individual translations, linear in the calibration
    parameters ``c_*`` (interface of ``boxtree.cost.FMMTranslationCostModel``,
    cost.py:87-150: ``direct()``, ``p2l/l2p/p2m/m2p(level)``,
    ``m2m/l2l/m2l(src_level, tgt_level)``, ``e2e_cost``)."""

    def __init__(self, ncoeffs_fmm_by_level, uses_point_and_shoot):
        self.ncoeffs_fmm_by_level = ncoeffs_fmm_by_level
        self.uses_point_and_shoot = uses_point_and_shoot

    @staticmethod
    def direct():
        return var("c_p2p")

    def e2e_cost(self, nsource_coeffs, ntarget_coeffs):
        if not self.uses_point_and_shoot:
            return nsource_coeffs * ntarget_coeffs
        # rotation to the z axis, translation along it, rotation back (cost.py:136-145)
        legs = ((nsource_coeffs, 3 / 2, 1), (nsource_coeffs, 1 / 2, ntarget_coeffs),
                (ntarget_coeffs, 3 / 2, 1))
        return sum(base ** power * factor for base, power, factor in legs)


for _name in _PARTICLE_OPERATORS:
    setattr(FMMTranslationCostModel, _name, _level_operator(_name))
for _name in _EXPANSION_OPERATORS:
    setattr(FMMTranslationCostModel, _name, _pair_operator(_name))
del _name


def make_pde_aware_translation_cost_model(dim, nlevels):
    """Expansions that use the PDE: (p+1)^(d-1) coefficients, point-and-shoot in 3D
    (cost.py:152-166)."""
    ncoeffs = [(var(f"p_fmm_lev{i}") + 1) ** (dim - 1) for i in range(nlevels)]
    return FMMTranslationCostModel(ncoeffs_fmm_by_level=ncoeffs, uses_point_and_shoot=dim == 3)


def make_taylor_translation_cost_model(dim, nlevels):
    """Cartesian Taylor expansions: (p+1)^d coefficients (cost.py:169-180)."""
    ncoeffs = [(var(f"p_fmm_lev{i}") + 1) ** dim for i in range(nlevels)]
    return FMMTranslationCostModel(ncoeffs_fmm_by_level=ncoeffs, uses_point_and_shoot=False)

# }}}


# {{{ cost model

class AbstractFMMCostModel:
    """Operation counts / calibrated costs of the FMM stages (cost.py:186-711)."""

    _FMM_STAGE_TO_CALIBRATION_PARAMETER = {
        "form_multipoles": "c_p2m", "coarsen_multipoles": "c_m2m", "eval_direct": "c_p2p",
        "multipole_to_local": "c_m2l", "eval_multipoles": "c_m2p", "form_locals": "c_p2l",
        "refine_locals": "c_l2l", "eval_locals": "c_l2p",
    }

    def __init__(self, translation_cost_model_factory=make_pde_aware_translation_cost_model):
        self.translation_cost_model_factory = translation_cost_model_factory

    @staticmethod
    def get_unit_calibration_params():
        return {"c_l2l": 1.0, "c_l2p": 1.0, "c_m2l": 1.0, "c_m2m": 1.0,
                "c_m2p": 1.0, "c_p2l": 1.0, "c_p2m": 1.0, "c_p2p": 1.0}

    @staticmethod
    def cost_factors_to_dev(cost_factors, actx):
        return {name: (actx.from_numpy(v) if isinstance(v, np.ndarray) and actx is not None else v)
                for name, v in cost_factors.items()}

    def fmm_cost_factors_for_kernels_from_model(self, actx, nlevels, xlat_cost, context):
        """Per-level translation costs for the ``process_*`` methods (cost.py:387-435)."""
        f64 = np.float64
        cost_factors = {
            "p2m_cost": np.array([evaluate(xlat_cost.p2m(lev), context) for lev in range(nlevels)], f64),
            "m2m_cost": np.array([evaluate(xlat_cost.m2m(lev + 1, lev), context)
                                  for lev in range(nlevels - 1)], f64),
            "c_p2p": evaluate(xlat_cost.direct(), context),
            "m2l_cost": np.array([evaluate(xlat_cost.m2l(lev, lev), context)
                                  for lev in range(nlevels)], f64),
            "m2p_cost": np.array([evaluate(xlat_cost.m2p(lev), context) for lev in range(nlevels)], f64),
            "p2l_cost": np.array([evaluate(xlat_cost.p2l(lev), context) for lev in range(nlevels)], f64),
            "l2l_cost": np.array([evaluate(xlat_cost.l2l(lev, lev + 1), context)
                                  for lev in range(nlevels - 1)], f64),
            "l2p_cost": np.array([evaluate(xlat_cost.l2p(lev), context) for lev in range(nlevels)], f64),
        }
        return self.cost_factors_to_dev(cost_factors, actx) if actx else cost_factors

    def _translation_cost(self, actx, traversal, level_to_order, calibration_params):
        tree = traversal.tree
        params = dict(calibration_params)
        for lev in range(int(tree.nlevels)):
            params[f"p_fmm_lev{lev}"] = level_to_order[lev]
            calibration_params[f"p_fmm_lev{lev}"] = level_to_order[lev]     # cost.py:472-473
        xlat_cost = self.translation_cost_model_factory(int(tree.dimensions), int(tree.nlevels))
        return self.fmm_cost_factors_for_kernels_from_model(actx, int(tree.nlevels), xlat_cost, params)

    def cost_per_box(self, actx, traversal, level_to_order, calibration_params,
                     ndirect_sources_per_target_box=None, box_target_counts_nonchild=None):
        """[nboxes] cost of all stages for each box (cost.py:445-525)."""
        if ndirect_sources_per_target_box is None:
            ndirect_sources_per_target_box = self.get_ndirect_sources_per_target_box(actx, traversal)
        tree = traversal.tree
        tc = self._translation_cost(actx, traversal, level_to_order, calibration_params)
        if box_target_counts_nonchild is None:
            box_target_counts_nonchild = tree.box_target_counts_nonchild
        result = self.zero_cost_per_box(actx, int(tree.nboxes))
        sb = traversal.source_boxes.long()
        tb = traversal.target_boxes.long()
        ttp = traversal.target_or_target_parent_boxes.long()
        result[sb] += self.process_form_multipoles(actx, traversal, tc["p2m_cost"])
        result[tb] += self.process_direct(actx, traversal, ndirect_sources_per_target_box,
                                          tc["c_p2p"],
                                          box_target_counts_nonchild=box_target_counts_nonchild)
        result[ttp] += self.process_list2(actx, traversal, tc["m2l_cost"])
        result += self.process_list3(actx, traversal, tc["m2p_cost"],
                                     box_target_counts_nonchild=box_target_counts_nonchild)
        result[ttp] += self.process_list4(actx, traversal, tc["p2l_cost"])
        result[tb] += self.process_eval_locals(actx, traversal, tc["l2p_cost"],
                                               box_target_counts_nonchild=box_target_counts_nonchild)
        return result

    def cost_per_stage(self, actx, traversal, level_to_order, calibration_params,
                       ndirect_sources_per_target_box=None, box_target_counts_nonchild=None):
        """Stage name -> cost (cost.py:527-624)."""
        if ndirect_sources_per_target_box is None:
            ndirect_sources_per_target_box = self.get_ndirect_sources_per_target_box(actx, traversal)
        tree = traversal.tree
        tc = self._translation_cost(actx, traversal, level_to_order, calibration_params)
        if box_target_counts_nonchild is None:
            box_target_counts_nonchild = tree.box_target_counts_nonchild
        agg = self.aggregate_over_boxes
        return {
            "form_multipoles": agg(actx, self.process_form_multipoles(actx, traversal, tc["p2m_cost"])),
            "coarsen_multipoles": self.process_coarsen_multipoles(actx, traversal, tc["m2m_cost"]),
            "eval_direct": agg(actx, self.process_direct(
                actx, traversal, ndirect_sources_per_target_box, tc["c_p2p"],
                box_target_counts_nonchild=box_target_counts_nonchild)),
            "multipole_to_local": agg(actx, self.process_list2(actx, traversal, tc["m2l_cost"])),
            "eval_multipoles": agg(actx, self.process_list3(
                actx, traversal, tc["m2p_cost"],
                box_target_counts_nonchild=box_target_counts_nonchild)),
            "form_locals": agg(actx, self.process_list4(actx, traversal, tc["p2l_cost"])),
            "refine_locals": self.process_refine_locals(actx, traversal, tc["l2l_cost"]),
            "eval_locals": agg(actx, self.process_eval_locals(
                actx, traversal, tc["l2p_cost"],
                box_target_counts_nonchild=box_target_counts_nonchild)),
        }

    def estimate_calibration_params(self, model_results, timing_results,
                                    time_field_name="wall_elapsed",
                                    additional_stage_to_param_names=()):
        """Least-squares fit, one parameter per stage, of measured times against
        unit-calibration model results (cost.py:650-709)."""
        nresults = len(model_results)
        assert len(timing_results) == nresults
        stage_to_param = dict(self._FMM_STAGE_TO_CALIBRATION_PARAMETER)
        stage_to_param.update(additional_stage_to_param_names)
        params = set(stage_to_param.values())
        modeled = {p: np.zeros(nresults) for p in params}
        measured = {p: np.zeros(nresults) for p in params}
        for icase, model_result in enumerate(model_results):
            for stage, param in stage_to_param.items():
                if stage in model_result:
                    modeled[param][icase] = float(model_result[stage])
        for icase, timing_result in enumerate(timing_results):
            for stage, time in timing_result.items():
                measured[stage_to_param[stage]][icase] = time[time_field_name]
        result = {}
        for param in params:
            m, t = modeled[param], measured[param]
            result[param] = 0.0 if np.allclose(m, 0) else t.dot(m) / m.dot(m)
        return result


class FMMCostModel(AbstractFMMCostModel):
    """The per-box processors on the device (cost.py:715-1260)."""

    def zero_cost_per_box(self, actx, nboxes):
        return actx.zeros(nboxes, np.float64)

    def aggregate_over_boxes(self, actx, per_box_result):
        if isinstance(per_box_result, float):
            return per_box_result
        return float(per_box_result.sum())

    @staticmethod
    def _csr_rows(actx, starts, lists, box_values):
        nrows = int(starts.shape[0]) - 1
        out = actx.zeros(nrows, np.float64)
        actx.sync_in()
        _lib.check(actx.lib.bt_fmm_csr_sum(actx.handle, nrows, ptr(starts), ptr(lists.contiguous()),
                                           ptr(box_values), None, ptr(out), 0))
        return out

    @staticmethod
    def _level_ranges(actx, level_starts):
        return [int(v) for v in actx.to_numpy(level_starts)]

    def process_form_multipoles(self, actx, traversal, p2m_cost):
        tree = traversal.tree
        sb = traversal.source_boxes.long()
        return (tree.box_source_counts_nonchild[sb].double()
                * p2m_cost[tree.box_levels[sb].long()])                 # cost.py:1265-1277

    def process_coarsen_multipoles(self, actx, traversal, m2m_cost):
        tree = traversal.tree
        lev = self._level_ranges(actx, traversal.level_start_source_parent_box_nrs)
        m2m = actx.to_numpy(m2m_cost)
        nb = int(tree.nboxes)
        nchildren = (tree.box_child_ids[:, :nb] != 0).sum(dim=0)
        result = 0.0
        for source_level in range(int(tree.nlevels) - 1, 2, -1):       # cost.py:1397-1412
            target_level = source_level - 1
            boxes = traversal.source_parent_boxes[lev[target_level]:lev[target_level + 1]].long()
            result += float(m2m[target_level]) * int(nchildren[boxes].sum())
        return result

    def get_ndirect_sources_per_target_box(self, actx, traversal):
        tree = traversal.tree
        counts = tree.box_source_counts_nonchild.double()
        n = self._csr_rows(actx, traversal.neighbor_source_boxes_starts,
                           traversal.neighbor_source_boxes_lists, counts)
        if traversal.from_sep_close_smaller_starts is not None:         # cost.py:1295-1301
            n = n + self._csr_rows(actx, traversal.from_sep_close_smaller_starts,
                                   traversal.from_sep_close_smaller_lists, counts)
        if traversal.from_sep_close_bigger_starts is not None:
            n = n + self._csr_rows(actx, traversal.from_sep_close_bigger_starts,
                                   traversal.from_sep_close_bigger_lists, counts)
        return n

    def process_direct(self, actx, traversal, ndirect_sources_by_itgt_box, p2p_cost,
                       box_target_counts_nonchild=None):
        if box_target_counts_nonchild is None:
            box_target_counts_nonchild = traversal.tree.box_target_counts_nonchild
        ntargets = box_target_counts_nonchild[traversal.target_boxes.long()].double()
        return ntargets * ndirect_sources_by_itgt_box * p2p_cost        # cost.py:1314-1322

    def process_list2(self, actx, traversal, m2l_cost):
        tree = traversal.tree
        ttp = traversal.target_or_target_parent_boxes.long()
        starts = traversal.from_sep_siblings_starts
        return m2l_cost[tree.box_levels[ttp].long()] * (starts[1:] - starts[:-1]).double()

    def process_list3(self, actx, traversal, m2p_cost, box_target_counts_nonchild=None):
        tree = traversal.tree
        if box_target_counts_nonchild is None:
            box_target_counts_nonchild = tree.box_target_counts_nonchild
        nm2p = self.zero_cost_per_box(actx, int(tree.nboxes))
        for ilevel, ssn in enumerate(traversal.from_sep_smaller_by_level):   # cost.py:1346-1352
            tboxes = traversal.target_boxes_sep_smaller_by_source_level[ilevel].long()
            if int(tboxes.shape[0]) == 0:
                continue
            nlist = (ssn.starts[1:] - ssn.starts[:-1]).double()
            nm2p[tboxes] += box_target_counts_nonchild[tboxes].double() * nlist * m2p_cost[ilevel]
        return nm2p

    def process_list4(self, actx, traversal, p2l_cost):
        tree = traversal.tree
        per_source_box = (tree.box_source_counts_nonchild.double()
                          * p2l_cost[tree.box_levels.long()])            # cost.py:1362-1365
        return self._csr_rows(actx, traversal.from_sep_bigger_starts,
                              traversal.from_sep_bigger_lists, per_source_box)

    def process_eval_locals(self, actx, traversal, l2p_cost, box_target_counts_nonchild=None):
        tree = traversal.tree
        if box_target_counts_nonchild is None:
            box_target_counts_nonchild = tree.box_target_counts_nonchild
        tb = traversal.target_boxes.long()
        return box_target_counts_nonchild[tb].double() * l2p_cost[tree.box_levels[tb].long()]

    def process_refine_locals(self, actx, traversal, l2l_cost):
        tree = traversal.tree
        lev = self._level_ranges(actx, traversal.level_start_target_or_target_parent_box_nrs)
        l2l = actx.to_numpy(l2l_cost)
        result = 0.0
        for target_lev in range(1, int(tree.nlevels)):                  # cost.py:1417-1422
            result += (lev[target_lev + 1] - lev[target_lev]) * float(l2l[target_lev - 1])
        return result

# }}}
