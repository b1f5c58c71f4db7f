#!/bin/bash
# stand-alone products with parts of the 4x4x4 kernel compiled out (library built with -DVMX_GEMM_DIAG; results are
# wrong on purpose: timing only)
export VMX_GEMM_44=1
for ab in 0 1 3 4 7; do
  echo "ablate=$ab"; VMX_GEMM_ABLATE=$ab python3 scripts/gpu_gemm_sweep.py 256 2>/dev/null | grep "n=5000\|n=2500"
done
