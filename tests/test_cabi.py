"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
every symbol include/boxtree_hip.h declares (no compute without a GPU)."""

import os
import re

from boxtree_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    import __graft_entry__ as g
    g.build()
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "boxtree_hip.h")).read()
    declared = set(re.findall(r"\b(bt_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.bt_abi_version() == _lib.ABI_VERSION


def test_no_gpu_fails_loudly():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from boxtree_amd import HIPArrayContext
    with pytest.raises(RuntimeError):
        HIPArrayContext()


def test_struct_sizes_match_header():
    # guards against ctypes/C layout drift: sizes computed from the header's
    # field lists with natural alignment
    import ctypes as ct
    assert ct.sizeof(_lib.TreeParams) == 8 + 16 + 24 + 24 + 8 * 3 + 16 + 8 + 48 + 8 + 16 + 16 + 16 + 16
    assert ct.sizeof(_lib.TreeSizes) == 16 + 8 + 4 * 65 + 4 + 7 * 8
    assert ct.sizeof(_lib.TravSizes) == 8 * 10 + 8 * 64 * 2


def test_struct_layouts_match_the_header(tmp_path):
    """Size and the offset of every field of each ctypes mirror in boxtree_amd/_lib.py
    equal what a C compiler lays out for include/boxtree_hip.h."""
    import ctypes as ct
    import subprocess

    from boxtree_amd import _lib
    pairs = {
        "bt_sort_stats": _lib.SortStats, "bt_tree_params": _lib.TreeParams,
        "bt_tree_sizes": _lib.TreeSizes, "bt_tree_arrays": _lib.TreeArrays,
        "bt_stage_times": _lib.StageTimes, "bt_trav_params": _lib.TravParams,
        "bt_trav_sizes": _lib.TravSizes, "bt_trav_arrays": _lib.TravArrays,
        "bt_aq_tree": _lib.AqTree, "bt_trav_packed": _lib.TravPacked,
        "bt_mgpu_params": _lib.MgpuParams, "bt_mgpu_shard": _lib.MgpuShard,
    }
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "boxtree_hip.h"',
             'int main(void) {']
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(root, "include"), str(src),
                           "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    for line in out.splitlines():
        cname, field, value = line.split()
        cls = pairs[cname]
        if field == "size":
            assert ct.sizeof(cls) == int(value), (cname, ct.sizeof(cls), value)
        else:
            assert getattr(cls, field).offset == int(value), (cname, field)


def test_mgpu_plan_matches_the_python_plan():
    """bt_mgpu_plan (host part of bt_mgpu_exchange: owners of the level-k cells, leaves of
    the global top tree kept whole) against the numpy statement used by the
    torch.distributed path (boxtree_amd/distributed: top_tree_plan + partition_cells).
    Pure host code: runs without a GPU."""
    import ctypes as ct

    import numpy as np

    from boxtree_amd.distributed import partition_cells, top_tree_plan
    lib = _lib.load()
    rng = np.random.default_rng(3)
    for dims, k in ((3, 3), (3, 4), (2, 5), (1, 7)):
        ncells = 1 << (dims * k)
        for trial in range(4):
            # clustered counts: a few heavy cells, many empty ones
            hist = (rng.random(ncells) < 0.3) * rng.integers(0, 200, ncells)
            hist[rng.integers(0, ncells, 3)] += rng.integers(1000, 100000, 3)
            hist = hist.astype(np.int64)
            for world in (1, 2, 5, 8):
                for mpb in (0, 30, 5000):
                    owner = np.empty(ncells, np.int32)
                    prefix = np.empty(ncells + 1, np.int64)
                    code = lib.bt_mgpu_plan(dims, k, mpb, world, hist.ctypes.data_as(ct.c_void_p),
                                            owner.ctypes.data_as(ct.c_void_p),
                                            prefix.ctypes.data_as(ct.c_void_p))
                    assert code == 0
                    if mpb > 0:
                        plan = top_tree_plan(hist, dims, k, mpb)
                        want = partition_cells(hist, world, plan["unit_start"])
                        assert np.array_equal(prefix, plan["cell_prefix"])
                    else:
                        want = partition_cells(hist, world, None)
                    assert np.array_equal(owner, want), (dims, k, world, mpb)


def test_mgpu_plan_with_extents_from_particles():
    """bt_mgpu_plan_ext (the plan of a sharded build whose targets have extents) against a
    particle-level statement: particles are (level-k cell, stop level); a box's arrivals are the
    particles under it that did not stop above it, it splits iff those that do not stop IN it
    exceed the limit (tree_build_kernels.py:569-591), a child exists iff its parent splits and
    something arrives; leaves above level k go to one rank; a particle that stays in a box counts
    for the box's first cell when the owners are balanced.  Pure host code."""
    import ctypes as ct

    import numpy as np

    from boxtree_amd.distributed import partition_cells
    lib = _lib.load()
    rng = np.random.default_rng(17)
    for dims, k in ((3, 3), (2, 4), (1, 6), (3, 2)):
        C = 1 << dims
        ncells = C ** k
        off = [(C ** lev - 1) // (C - 1) for lev in range(k + 2)]
        for trial in range(4):
            n = int(rng.integers(2000, 20000))
            # clustered cells; most particles go below level k, some stop at a level <= k
            cell = np.where(rng.random(n) < 0.5, rng.integers(0, ncells, n),
                            rng.integers(0, max(ncells // 16, 1), n)).astype(np.int64)
            cap = np.where(rng.random(n) < 0.15, rng.integers(0, k + 1, n), k + 1)
            for mpb in (5, 60):
                # the tables the exchange makes: stayers per box, particles per (effective) cell
                stay = np.zeros(off[k + 1], np.int64)
                eff = cell.copy()
                for lev in range(k + 1):
                    m = cap == lev
                    sh = dims * (k - lev)
                    np.add.at(stay, off[lev] + (cell[m] >> sh), 1)
                    eff[m] = (cell[m] >> sh) << sh
                hist = np.bincount(eff, minlength=ncells).astype(np.int64)
                # particle-level statement
                arrive, split, exists = {}, {}, {0: np.ones(1, bool)}
                for lev in range(k + 1):
                    sh = dims * (k - lev)
                    m = cap >= lev
                    arrive[lev] = np.bincount(cell[m] >> sh, minlength=C ** lev)
                    desc = np.bincount(cell[cap > lev] >> sh, minlength=C ** lev)
                    split[lev] = exists[lev] & (desc > mpb)
                    if lev < k:
                        a_next = np.bincount(cell[cap >= lev + 1] >> (dims * (k - lev - 1)), minlength=C ** (lev + 1))
                        exists[lev + 1] = np.repeat(split[lev], C) & (a_next > 0)
                unit = np.arange(ncells)
                for lev in range(k - 1, -1, -1):          # the topmost box that does not split wins
                    sh = dims * (k - lev)
                    leafy = ~split[lev][np.arange(ncells) >> sh]
                    unit = np.where(leafy, (np.arange(ncells) >> sh) << sh, unit)
                for world in (1, 3, 8):
                    owner = np.empty(ncells, np.int32)
                    prefix = np.empty(ncells + 1, np.int64)
                    box_arrive = np.empty(off[k + 1], np.int64)
                    box_split = np.empty(off[k + 1], np.uint8)
                    vp = lambda a: a.ctypes.data_as(ct.c_void_p)      # noqa: E731
                    code = lib.bt_mgpu_plan_ext(dims, k, mpb, world, vp(hist), vp(stay), vp(owner),
                                                vp(prefix), vp(box_arrive), vp(box_split))
                    assert code == 0
                    for lev in range(k + 1):
                        sl = slice(off[lev], off[lev + 1])
                        assert np.array_equal(box_arrive[sl], arrive[lev]), (dims, k, lev)
                        assert np.array_equal((box_split[sl] & 1) != 0, exists[lev]), (dims, k, lev)
                        assert np.array_equal((box_split[sl] & 2) != 0, split[lev]), (dims, k, lev)
                    assert np.array_equal(prefix, np.concatenate([[0], np.cumsum(hist)]))
                    assert np.array_equal(owner, partition_cells(hist, world, unit)), (dims, k, world)


def test_lazy_a2a_time_behaves_like_a_number():
    import pytest
    """stats["a2a_ms"] of an exchange on a stream-ordered context resolves on first use and can be
    formatted, compared and used in arithmetic like the float it stands for; after another exchange
    on the same context it refuses to report that one's time."""
    import json

    from boxtree_amd.distributed import native as nat

    class FakeActx:
        _mgpu_exchange_serial = 3

    actx = FakeActx()
    calls = []
    orig = nat.exchange_time_ms
    nat.exchange_time_ms = lambda a: calls.append(a) or 1.25
    try:
        t = nat._LazyA2aTime(actx, 3)
        assert calls == []
        assert f"{t:.2f}" == "1.25" and calls == [actx]
        assert t + 1 == 2.25 and 1 + t == 2.25 and t * 2 == 2.5 and t / 5 == 0.25 and 5 / t == 4.0
        assert t - 0.25 == 1.0 and 2 - t == 0.75 and -t == -1.25
        assert t > 1 and t >= 1.25 and t < 2 and t <= 1.25 and t == 1.25 and bool(t)
        assert round(t, 1) == 1.2 and json.dumps(float(t)) == "1.25" and max(t, 0.5) == 1.25
        assert len(calls) == 1                      # resolved once
        stale = nat._LazyA2aTime(actx, 2)           # an exchange before the context's last one
        with pytest.raises(RuntimeError, match="before the next exchange"):
            float(stale)
    finally:
        nat.exchange_time_ms = orig
