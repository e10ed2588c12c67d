#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_s7; mkdir -p $O
export TMPDIR=/tmp
DEV=$GRAFT_REPO_ROOT/tools/r6/libagp_dev.so
( timeout 600 python -m pytest tests/test_gpu_round6.py -x -q -m gpu 2>&1 | tail -8 ) > $O/tests_r6.txt
# (a) which wait expires at the C3 soak's shape (debug build), up to 12 runs of 20000 steps
( AGP_HIP_LIB=$DEV timeout 900 python tools/stress/split_which_wait.py 12 2048 2048 32 20000 2>&1 | grep -v amdgpu.ids ) > $O/which_wait_c3.txt
# (b) forced aborts with the diagnostic: where do the 4 s go
for i in 1 2 3; do
  ( AGP_HIP_LIB=$DEV AGP_STRESS_DIAG=1 AGP_CHAIN_SPLIT=1 AGP_DAG_TEST_ABORT=1 timeout 300 python tools/stress/abort_stress.py 60 2>&1 | grep -E "^model|^bad" | awk '{ if ($3+0 > 1.0 || /bad/) print }' ) >> $O/stressA_diag.txt
  echo "--- process $i done" >> $O/stressA_diag.txt
done
# (c) priorities: the ELBO side stream, the look-ahead at C3
for pr in none -1 0; do
  ( [ $pr = none ] || export AGP_BENCH_SIDE_PRIORITY=$pr; timeout 600 python bench.py --no-cpu-baseline --no-extras 2>$O/bench_c2_side_$pr.err | tail -1 ) > $O/bench_c2_side_$pr.json
done
for pr in lo no hi; do
  ( AGP_HIP_LIB=$DEV AGP_DEV_PF_PRIO=$pr timeout 600 python bench.py --config c3 --steps 200 --warmup 20 --no-cpu-baseline --no-elbo-tol --no-extras 2>$O/bench_c3_pf_$pr.err | tail -1 ) > $O/bench_c3_pf_$pr.json
done
