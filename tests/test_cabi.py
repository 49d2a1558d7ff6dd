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
    assert ct.sizeof(_lib.TreeParams) == 8 + 16 + 24 + 24 + 8 * 3 + 16 + 8 + 48 + 8 + 16 + 16
    assert ct.sizeof(_lib.TreeSizes) == 16 + 8 + 4 * 65 + 4
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
        "bt_aq_tree": _lib.AqTree,
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
