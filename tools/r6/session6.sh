#!/bin/bash
# round 6: the full GPU suite on the default build, then the big stress loops and the determinism soaks of the repaired split launch
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_s6; mkdir -p $O
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -30 ) > $O/gpu_suite.txt
# stress B: 1000 runs x 200 launches, default (fp32 launches with the prologue split) and split without the prologue
( timeout 1500 python tools/stress/split_hash_stress.py 1000 2>&1 | tail -2 ) > $O/stressB_default_1000.txt
( AGP_STEP_PROLOGUE=0 timeout 1500 python tools/stress/split_hash_stress.py 1000 2>&1 | tail -2 ) > $O/stressB_split_nopro_1000.txt
# stress A: 300 models of forced aborts, split with the prologue (5 processes of 60: a process that creates many contexts slows down)
for i in 1 2 3 4 5; do
  ( AGP_CHAIN_SPLIT=1 AGP_DAG_TEST_ABORT=1 timeout 300 python tools/stress/abort_stress.py 60 2>&1 | grep -E "^model|^bad" | awk '/^model/{n++; if ($3+0 > mx) mx=$3+0} /^bad/{print "models", n, "slowest", mx, "s", $0}' ) >> $O/stressA_split_pro_300.txt
done
# soaks
{ echo "# build of commit $(cat .git_head 2>/dev/null)";
  echo "## C2 shape fp64, 2 x 60000 steps (merged launch, look-ahead on)"; timeout 900 python tools/soak_determinism.py 60000;
  echo "## C3 shape fp32 m = B = 2048, 2 x 20000 steps (split launch)"; timeout 900 python tools/soak_determinism.py 20000 2048 f32;
  echo "## fp32 m = 1024, B = 2048: split launch WITH the prologue (default again), 2 x 20000 steps"; timeout 900 python tools/soak_determinism.py 20000 1024 f32 cavi 2048;
  echo "## 8 latents (C4 shape), 2 x 5000 steps"; timeout 900 python tools/soak_multilatent.py 8 5000 2;
  echo "## hyper-on iteration, 2 x 2000"; timeout 900 python tools/soak_hyper.py 2000; } 2>&1 | grep -v amdgpu.ids > $O/soaks.txt
