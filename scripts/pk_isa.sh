#!/bin/bash
# dumps the gfx950 assembly of one kernel (name pattern $1) to $2
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$(mktemp -d /tmp/vmx_isa.XXXXXX)
cd "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC --save-temps -c -o v.o "$REPO/vega_amd/csrc/vegamx.hip" ${VMX_DEFS} 2>&1 | grep -i "error" || true
awk -v pat="$1" '$0 ~ "^_Z.*" pat ".*:$" {p=1} p {print} p && /s_endpgm/ {exit}' ./*gfx950*.s > "$2"
wc -l "$2"
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|sgpr_count|name):" ./*gfx950*.s | paste - - - - | grep -E "$1" | sed 's/ \+/ /g' | cut -c1-260
