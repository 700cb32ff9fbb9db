#!/bin/bash
# Developer tool: builds libamc.so once per "name:extra hipcc flags" argument and snapshots the package under
# tools/_variants/<name> for tools/gpu_ab.py (same-box A/B timing); restores the plain build at the end.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
[ -n "$KEEP" ] || rm -rf "$R/tools/_variants"
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}
  make -C "$R/montecarlo_amd/csrc" -B HIPFLAGS="$BASE $f" 2>&1 | grep -E " error|Error" || true
  python3 "$R/tools/gpu_ab.py" snapshot "$n"
done
make -C "$R/montecarlo_amd/csrc" -B 2>&1 | grep -E " error|Error" || true
