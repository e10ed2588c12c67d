export TMPDIR=/tmp
R=$PWD
B="python $R/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-elbo-tol --no-extras"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline'].get('avg_launch_us'))" "$1"; }
$B 2>/dev/null | line "c2 epi"
AGP_STEP_EPILOGUE=0 $B 2>/dev/null | line "c2 no-epi"
(cd /tmp && AGP_BENCH_NO_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_c2 -o p -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-elbo-tol --no-extras > /dev/null 2>&1)
python tools/step_gaps.py gpurun_out/prof_r03_c2/p_kernel_trace.csv | tee gpurun_out/r03_c2_step_gaps.txt
