python -m pytest tests -m gpu -x -q 2>&1 | grep "passed\|failed" | tail -1
for c in c2 c4 c4; do
python bench.py --config $c --no-extras --no-cpu-baseline --no-elbo-tol --steps 200 --warmup 20 2>/dev/null < /dev/null | tail -1 > gpurun_out/b.json
python - <<PY
import json
d=json.loads(open("gpurun_out/b.json").read())
print("$c", d["ms_per_step"], d["value"])
PY
done
