#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_s5; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kernelmatrix or potrf or gemm or cavi_trajectory or large_m1024" 2>&1 | tail -15 ) > $O/tests_quick.txt
for c in c2 c4 c5 c3; do
  ( timeout 900 python bench.py --config $c --steps 40 --warmup 5 --no-cpu-baseline --no-elbo-tol 2>$O/bench_$c.err | tail -1 ) > $O/bench_$c.json
done
( timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -40 ) > $O/gpu_suite.txt
