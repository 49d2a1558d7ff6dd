"""A cross-section of the `-m gpu` parity tests against the CPU emulation of the kernels
(tests/emu/README.md), in the CPU suite: the kernels of boxtree_amd/csrc, compiled unmodified for
the host, build trees and interaction lists that equal the oracle's.  Test infrastructure for boxes
without a GPU -- the kernels' logic, nothing the hardware decides.  Runs in a subprocess: a process
uses either the product library or the emulated one.

The whole suite under emulation: tools/emu_suite.sh (log: profiles/r06_emu_pytest.txt)."""

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (small cases of every stage: sorts, packed / pair / continuation keys, extents, level restriction,
# the three list paths, walk rows with spill and overflow, both colleague-row families, sharded
# builds with thread ranks over the library's local communicator)
SELECTION = [
    "tests/test_gpu_parity.py::test_radix_sort_u64_keys[bits0-8193]",
    "tests/test_gpu_parity.py::test_radix_sort_u64[bits2-100003]",
    "tests/test_gpu_parity.py::test_packed_key_build_modes[3-7-2-pairs]",
    "tests/test_gpu_parity.py::test_walk_rows_overflow[2-30000-6-2-2-1]",
    "tests/test_gpu_parity.py::test_walk_rows_overflow[2-30000-6-64-3-0]",
    "tests/test_gpu_parity.py::test_colleague_row_families",
    "tests/test_gpu_parity.py::test_extent_tree",
    "tests/test_gpu_parity.py::test_tree_connectivity[2-True]",
    "tests/test_gpu_parity.py::test_deep_tree_below_the_key_with_extents",
    "tests/test_gpu_parity.py::test_multi_rank_native_entries[2-2-normal-1]",
    "tests/test_gpu_parity.py::test_multi_rank_native_entries[3-3-sphere-1]",
    "tests/test_gpu_parity.py::test_multi_rank_native_entries_extents[2-2-normal-1-l2]",
    "tests/test_gpu_level_restricted.py::test_level_restricted_tree",
    "tests/test_gpu_level_restricted.py::test_level_restricted_targets_extents_weights",
    "tests/test_gpu_area_query.py::test_peer_lists",
    "tests/test_gpu_area_query.py::test_area_query_reference_sizes",
    "tests/test_gpu_filters.py",
]


def emu_env():
    env = dict(os.environ, BOXTREE_EMU="1")
    env.pop("PYTEST_XDIST_WORKER", None)
    return env


def test_emulated_library_builds():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emu"), "-j8"],
                          stdout=subprocess.DEVNULL)
    import ctypes
    emu = ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "_build", "libboxtree_emu.so"))
    from boxtree_amd import _lib
    for name in _lib.EXPORTED_SYMBOLS:          # the same C ABI as the product library
        assert hasattr(emu, name), name


def test_emulator_known_answers():
    """The emulator itself against the documented semantics of what it emulates (shuffles, ballots,
    DPP controls, mbcnt, bitop3), the lock step of a wave's LDS traffic, a wave-wide operation under
    a branch the host compiler would like to clone, tickets / spins / dynamic LDS -- in the three
    orders the scheduler can give waves and lanes their turns in."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emu"), "-j8"], stdout=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "tests", "emu", "_build", "emu_selftest")
    for order in (None, "reverse", "shuffle:5"):
        env = dict(os.environ)
        env.pop("EMU_ORDER", None)
        if order:
            env["EMU_ORDER"] = order
        p = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
        assert p.returncode == 0 and "emu_selftest: ok" in p.stdout, (order, p.stdout[-2000:], p.stderr[-2000:])


def test_parity_cross_section_under_emulation():
    """About a minute on 8 cores; the whole suite: tools/emu_suite.sh (16 minutes, 518 tests)."""
    workers = str(max(1, min(6, len(os.sched_getaffinity(0)) - 1)))
    cmd = [sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-n", workers, "--timeout", "600",
           "-p", "no:cacheprovider", *SELECTION]
    p = subprocess.run(cmd, cwd=ROOT, env=emu_env(), capture_output=True, text=True, timeout=3000)
    tail = p.stdout[-3000:] + p.stderr[-2000:]
    assert p.returncode == 0, tail
    assert " passed" in p.stdout and "failed" not in p.stdout.splitlines()[-1], tail
