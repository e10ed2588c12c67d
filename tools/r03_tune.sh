export TMPDIR=/tmp
R=$PWD
(cd /tmp && AGP_BENCH_NO_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_c2 -o p -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-elbo-tol --no-extras > /dev/null 2>&1)
python tools/step_gaps.py gpurun_out/prof_r03_c2/p_kernel_trace.csv | tee gpurun_out/r03_c2_step_gaps.txt
(cd /tmp && AGP_PF_POLL=0 AGP_BENCH_NO_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_c2b -o p -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-elbo-tol --no-extras > /dev/null 2>&1)
python tools/step_gaps.py gpurun_out/prof_r03_c2b/p_kernel_trace.csv
