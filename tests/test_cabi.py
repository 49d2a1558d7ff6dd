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
    assert lib.bt_abi_version() == 1


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
