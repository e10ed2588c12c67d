#!/bin/bash
# round-5 regression pass on the GPU box: the GPU suite, then short bench lines of the four configs
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r5_pytest.log
cat gpurun_out/r5_pytest.log
for c in c2 c3 c4; do
  python bench.py --config $c --no-cpu-baseline --no-elbo-tol --steps 200 --warmup 30 2> gpurun_out/r5_${c}.err | tail -1 > gpurun_out/r5_${c}.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r5_${c}.json"))
print("${c}", d["ms_per_step"], d.get("ms_per_step_with_hyper_update"), d["roofline"]["avg_launch_us"])
PY
done
