#!/bin/bash
# The whole `-m gpu` suite against the CPU emulation of the kernels (tests/emu/README.md), log kept:
#   tools/emu_suite.sh [OUT=profiles/r06_emu_pytest.txt] [pytest args...]
# No GPU involved and nothing measured: which parity tests the kernels' LOGIC passes.  Tests that
# need the hardware itself (RCCL, the plain-C programs linked against the product library, bench.py,
# HIP streams) fail or skip here by construction; tests at BASELINE's full sizes run out of time.
set -u
cd "$(dirname "$0")/.."
OUT=${1:-profiles/r06_emu_pytest.txt}; shift || true
make -C tests/emu -j8 > /dev/null || exit 1
{
  echo "# BOXTREE_EMU=1 python -m pytest tests -m gpu -n 7 --timeout ${EMU_TIMEOUT:-600} $*"
  echo "# HEAD $(git rev-parse --short HEAD)  $(date -u +%Y-%m-%dT%H:%MZ)  host: $(nproc) cores, no GPU"
  BOXTREE_EMU=1 python -m pytest tests -m gpu -q -n 7 --timeout "${EMU_TIMEOUT:-600}" -p no:cacheprovider -rfEs "$@" 2>&1 \
    | grep -v "^\[gw\|^bringing up\|^$" | cut -c1-400
} > "$OUT"
tail -3 "$OUT"
