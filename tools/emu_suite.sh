#!/bin/bash
# The `-m gpu` suite against the CPU emulation of the kernels (tests/emu/README.md), log kept:
#   tools/emu_suite.sh [OUT=profiles/r06_emu_pytest.txt] [extra pytest args...]
# No GPU involved and nothing measured: which parity tests the kernels' LOGIC passes on a CPU.
# tests/emu/needs_hardware.txt lists what is left out (RCCL, HIP streams, programs linked against
# the product library, BASELINE's full sizes) with the reason per test.
set -u
cd "$(dirname "$0")/.."
OUT=${1:-profiles/r06_emu_pytest.txt}; shift || true
make -C tests/emu -j8 > /dev/null || exit 1
DESEL=()
while read -r id _; do
  case "$id" in ''|\#*) continue;; esac
  DESEL+=(--deselect "$id")
done < tests/emu/needs_hardware.txt
{
  echo "# BOXTREE_EMU=1 python -m pytest tests -m gpu -n 7 --timeout ${EMU_TIMEOUT:-900} (minus tests/emu/needs_hardware.txt) $*"
  echo "# switches in the environment: $(env | grep '^BT_\|^EMU_' | tr '\n' ' ')"
  echo "# HEAD $(git rev-parse --short HEAD)  $(date -u +%Y-%m-%dT%H:%MZ)  host: $(nproc) cores, no GPU"
  echo "# not run under emulation (tests/emu/needs_hardware.txt):"
  grep -v "^#" tests/emu/needs_hardware.txt | sed 's/^/#   /'
  BOXTREE_EMU=1 python -m pytest tests -m gpu -q -n 7 --timeout "${EMU_TIMEOUT:-900}" -p no:cacheprovider -rfEs \
      "${DESEL[@]}" "$@" 2>&1 | grep -v "^\[gw\|^bringing up\|^$" | cut -c1-400
} > "$OUT"
tail -3 "$OUT"
