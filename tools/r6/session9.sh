#!/bin/bash
# round 6, FINAL collection on the frozen sources: tools/collect_round6.sh (pmc, lines, stats, predict, soaks), then the GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r06_gpu_suite.txt
bash tools/collect_round6.sh pmc lines stats predict soaks > gpurun_out/r06_collect.log 2>&1
ls gpurun_out | grep r06_
