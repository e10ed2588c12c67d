python tools/bench_diag.py > gpurun_out/r02_diag.log 2>&1
for i in 1 2; do
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-elbo-tol --no-extras > gpurun_out/r02_b4_c2_def_$i.json 2>/dev/null
AGP_HIP_LIB=$PWD/augmentedgaussianprocesses.jl_amd/libagp_hip_piv1.so python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-elbo-tol --no-extras > gpurun_out/r02_b4_c2_piv1_$i.json 2>/dev/null
done
AGP_HIP_LIB=$PWD/augmentedgaussianprocesses.jl_amd/libagp_hip_piv1.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r02_t4_piv1.log
cat gpurun_out/r02_diag.log; tail -3 gpurun_out/r02_t4_piv1.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02_b4_c2_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["roofline"]["avg_launch_us"])
PY
