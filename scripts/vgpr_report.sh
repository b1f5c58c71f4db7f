#!/bin/bash
# Register report of the engine's kernels: compiles vegamx.hip for gfx950 with --save-temps in a scratch directory and
# prints name / VGPR count / spilled VGPRs of the kernels whose name matches $1 (default: all).
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$(mktemp -d /tmp/vmx_regs.XXXXXX)
cd "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC --save-temps -c -o v.o "$REPO/vega_amd/csrc/vegamx.hip" 2>&1 | grep -i "error" || true
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|name):" ./*gfx950*.s | paste - - - | grep -E "${1:-.}" | sed 's/ \+/ /g' | cut -c1-220
