#!/bin/bash
# A/B helper: tools/ab.sh <config> <steps> "ENV1=a ENV2=b" "ENV1=c" ...  -> ms_per_step, iter/s, dominant-kernel us, frac per variant
cfg=$1; steps=$2; shift 2
for envs in "$@"; do
  line=$(env $envs timeout 150 python bench.py --config $cfg --steps $steps --warmup 20 --no-cpu-baseline --no-elbo-tol --no-extras 2>/dev/null | tail -1)
  echo "$cfg [$envs] $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d.get("roofline",{}); print("ms/step", d["ms_per_step"], "it/s", d["value"], "kernel_us", r.get("avg_launch_us"), "frac", r.get("frac"))')"
done
