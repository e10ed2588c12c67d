#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_s2; mkdir -p $O
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 ) > $O/gpu_suite.txt
( timeout 900 python bench.py --no-cpu-baseline 2>$O/bench_c2.err | tail -1 ) > $O/bench_c2.json
