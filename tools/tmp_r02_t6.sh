python tools/bench_diag.py 2>&1 | grep "f64" > gpurun_out/r02_diag2.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_round2.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r02_t6.log
python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-elbo-tol > gpurun_out/r02_b6_c2.json 2>/dev/null
python bench.py --config c3 --steps 50 --warmup 10 --no-cpu-baseline --no-elbo-tol > gpurun_out/r02_b6_c3.json 2>/dev/null
cat gpurun_out/r02_diag2.log; tail -3 gpurun_out/r02_t6.log
python - <<'PY'
import json,glob
for f in ["gpurun_out/r02_b6_c2.json","gpurun_out/r02_b6_c3.json"]:
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["roofline"]["avg_launch_us"], d.get("predict_f_mean_all_N"))
PY
