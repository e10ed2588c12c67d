#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_s16; mkdir -p $O
for mode in lookahead nolookahead; do
  for i in 1 2 3 4 5 6; do
    ( [ $mode = nolookahead ] && export AGP_STRESS_NO_LOOKAHEAD=1; AGP_CHAIN_SPLIT=1 AGP_DAG_TEST_ABORT=1 timeout 300 python tools/stress/abort_stress.py 60 2>&1 | grep -E "^model|^bad" | awk '/^model/{n++; if ($3+0 > mx) mx=$3+0; if ($3+0 > 2.0) st++} /^bad/{print "models", n, "slowest", mx, "s, models over 2 s:", st+0, $0}' ) >> $O/stressA_$mode.txt
  done
done
( for i in 1 2 3; do AGP_CHAIN_SPLIT=1 AGP_STEP_PROLOGUE=0 AGP_DAG_TEST_ABORT=1 timeout 300 python tools/stress/abort_stress.py 60 2>&1 | grep -E "^model|^bad" | awk '/^model/{n++; if ($3+0 > mx) mx=$3+0; if ($3+0 > 2.0) st++} /^bad/{print "models", n, "slowest", mx, "s, models over 2 s:", st+0, $0}'; done ) > $O/stressA_split_nopro.txt
tail -n +1 $O/*.txt
