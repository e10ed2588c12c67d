#!/bin/bash
# round 6, first GPU session: the repaired split launch (DagSync::here, bounded grid barrier)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_split1; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "split" 2>&1 | tail -15 ) > $O/tests_split.txt
# stress B: 300 runs x 200 launches each, default (fp32 prologue split), split without prologue, merged
( timeout 600 python tools/stress/split_hash_stress.py 300 2>&1 | tail -3 ) > $O/stressB_default.txt
( AGP_STEP_PROLOGUE=0 timeout 600 python tools/stress/split_hash_stress.py 300 2>&1 | tail -3 ) > $O/stressB_split_nopro.txt
( AGP_CHAIN_SPLIT=0 timeout 400 python tools/stress/split_hash_stress.py 100 2>&1 | tail -3 ) > $O/stressB_merged.txt
# stress A: forced aborts, split with prologue (fp64 m = B = 1024)
( AGP_CHAIN_SPLIT=1 AGP_DAG_TEST_ABORT=1 timeout 300 python tools/stress/abort_stress.py 60 2>&1 | tail -70 ) > $O/stressA_split_pro.txt
( AGP_CHAIN_SPLIT=1 AGP_STEP_PROLOGUE=0 AGP_DAG_TEST_ABORT=1 timeout 300 python tools/stress/abort_stress.py 60 2>&1 | tail -70 ) > $O/stressA_split_nopro.txt
# step times with the wait kernel in front of the tile kernel
for c in c3 c4; do
  ( timeout 600 python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline --no-elbo-tol --no-extras 2>$O/bench_$c.err | tail -1 ) > $O/bench_$c.json
done
ls -la $O
