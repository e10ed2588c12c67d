#!/bin/bash
# residual rate of task-graph fallbacks at the C3 shape: 5 x (2 x 20000) steps
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2 3 4 5; do timeout 600 python tools/soak_determinism.py 20000 2048 f32 2>&1 | grep -E "^run|bitwise|warning" | cut -c1-200; done > gpurun_out/r06_c3_residual.txt
cat gpurun_out/r06_c3_residual.txt
