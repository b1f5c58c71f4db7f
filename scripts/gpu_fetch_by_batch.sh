for b in 64 128 256 512; do echo "B=$b"; BENCH_ARGS="--batch $b" bash scripts/gpu_fetch_compare.sh "VMX_QUAD_KBANDS=0" "VMX_QUAD_KBANDS=1"; done
