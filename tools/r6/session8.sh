#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_s8; mkdir -p $O
export TMPDIR=/tmp
DEV=$GRAFT_REPO_ROOT/tools/r6/libagp_dev.so
B="--no-cpu-baseline --no-elbo-tol --no-extras"
for pr in lo hi no lo hi; do
  ( AGP_HIP_LIB=$DEV AGP_DEV_PF_PRIO=$pr timeout 600 python bench.py --config c3 --steps 300 --warmup 30 $B 2>/dev/null | tail -1 ) >> $O/c3_$pr.json
done
for pr in lo hi; do
  ( AGP_HIP_LIB=$DEV AGP_DEV_PF_PRIO=$pr timeout 600 python bench.py --config c2 --steps 300 --warmup 30 $B 2>/dev/null | tail -1 ) >> $O/c2_$pr.json
  ( AGP_HIP_LIB=$DEV AGP_DEV_PF_PRIO=$pr timeout 600 python bench.py --config c2 --m 2048 --batch 2048 --steps 100 --warmup 10 $B 2>/dev/null | tail -1 ) >> $O/c2_m2048_$pr.json
  ( AGP_HIP_LIB=$DEV AGP_DEV_PF_PRIO=$pr timeout 600 python bench.py --config c3 --m 1024 --batch 2048 --steps 300 --warmup 30 $B 2>/dev/null | tail -1 ) >> $O/c3_m1024_b2048_$pr.json
done
( timeout 600 python -m pytest tests/test_gpu_round6.py -x -q -m gpu 2>&1 | tail -5 ) > $O/tests_r6.txt
