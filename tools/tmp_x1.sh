python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_comm.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | grep "passed\|failed\|error" | tail -3
for c in c2 c3 c4 c5; do
st=200; [ $c = c5 ] && st=20
python bench.py --config $c --no-extras --no-cpu-baseline --no-elbo-tol --steps $st --warmup 10 > gpurun_out/x1_$c.json 2>gpurun_out/x1_$c.err < /dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/x1_$c.json").read().strip().splitlines()[-1])
print("$c", d["ms_per_step"], d["value"])
PY
done
