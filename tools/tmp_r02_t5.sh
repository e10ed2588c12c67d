python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_round2.py tests/test_gpu_online.py tests/test_gpu_kmeans.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r02_t5.log
python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-elbo-tol > gpurun_out/r02_b5_c2.json 2>/dev/null
python bench.py --config c3 --steps 50 --warmup 10 --no-cpu-baseline --no-elbo-tol > gpurun_out/r02_b5_c3.json 2>/dev/null
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02_c3 -o p -- python $R/bench.py --config c3 --steps 50 --warmup 10 --no-cpu-baseline --no-elbo-tol > $R/gpurun_out/prof_r02_c3.line 2>/dev/null)
python tools/pmc_traffic.py collect c3 r02_c3 > /dev/null 2>&1; python tools/pmc_traffic.py parse c3 r02_c3 > gpurun_out/r02_pmc_c3.log 2>&1
tail -4 gpurun_out/r02_t5.log
python - <<'PY'
import json,glob
for f in ["gpurun_out/r02_b5_c2.json","gpurun_out/r02_b5_c3.json"]:
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["roofline"]["avg_launch_us"], d.get("predict_f_mean_all_N"))
PY
f=$(find gpurun_out/prof_r02_c3 -name "*kernel_stats.csv" | head -1); python tools/summarize_prof.py $f 16
cat gpurun_out/r02_pmc_c3.log | head -30
