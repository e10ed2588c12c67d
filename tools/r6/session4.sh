#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_s4; mkdir -p $O
export TMPDIR=/tmp
( timeout 120 ./tools/ubench/mfma4 > $O/mfma4.txt 2>&1 )
( timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -m gpu 2>&1 | tail -25 ) > $O/tests_r6.txt
for tall in 0 1; do
  for c in c5 c4; do
    ( AGP_GEMM_TALL=$tall timeout 900 python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline --no-elbo-tol --no-extras 2>$O/bench_${c}_tall$tall.err | tail -1 ) > $O/bench_${c}_tall$tall.json
  done
done
( AGP_GEMM_TALL=1 timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -x -q -m gpu -k "c5_path or c4_path or large_m1024 or trajectory or golden" 2>&1 | tail -5 ) > $O/tests_tall.txt
