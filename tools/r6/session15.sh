#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_s15; mkdir -p $O
for lag in 1 2 3 1 2; do
  ( AGP_BENCH_ELBO_LAG=$lag timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 ) >> $O/lag$lag.json
done
