#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_s11; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round4.py -x -q -m gpu 2>&1 | tail -6 ) > $O/tests.txt
( AGP_CHOL_DAG=0 timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -m gpu 2>&1 | tail -3 ) > $O/tests_nodag.txt
( timeout 900 python bench.py --no-cpu-baseline --no-extras 2>$O/bench_c2.err | tail -1 ) > $O/bench_c2.json
( AGP_BENCH_ELBO_INLINE=1 timeout 900 python bench.py --no-cpu-baseline --no-extras 2>$O/bench_c2_inline.err | tail -1 ) > $O/bench_c2_inline.json
