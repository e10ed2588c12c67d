#!/bin/bash
# the GPU suite once with each remaining fallback forced.  usage: bash tools/suite_with_fallbacks.sh [KNOB=VALUE ...]
# (default: every fallback-forcing knob of include/agp_hip.h "Environment").  Output: gpurun_out/r06_fallback_suites.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/r06_fallback_suites.txt
KNOBS=("$@")
[ ${#KNOBS[@]} -eq 0 ] && KNOBS=(AGP_CHOL_DAG=0 AGP_CHAIN_SPLIT=1 AGP_CHAIN_SPLIT=0 AGP_STEP_PROLOGUE=0 AGP_STEP_EPILOGUE=0 AGP_PF_INKERNEL=0 AGP_KERNELMATRIX_VALU=1 AGP_HYPER_GK_FUSED=0 AGP_SPLIT_MERGED=0)
: > $OUT
for kv in "${KNOBS[@]}"; do
  echo "=== $kv" | tee -a $OUT
  env $kv timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --deselect tests/test_gpu_bench_smoke.py 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed" | cut -c1-260 | tee -a $OUT
done
