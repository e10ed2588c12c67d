export TMPDIR=/tmp
R=$PWD
T=r03
c=c5; st=20; wu=5
python bench.py --config $c --steps $st --warmup $wu --cpu-elbo-seconds 0 > gpurun_out/${T}_${c}_bench_line.json 2> gpurun_out/${T}_${c}_line.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${T}_$c -o p -- python $R/bench.py --config $c --steps $st --warmup $wu --no-cpu-baseline --no-elbo-tol --no-extras > /dev/null 2>&1)
cp gpurun_out/prof_${T}_$c/p_kernel_stats.csv gpurun_out/${T}_${c}_kernel_stats.csv
python tools/summarize_prof.py gpurun_out/${T}_${c}_kernel_stats.csv 8; tail -c 900 gpurun_out/${T}_${c}_bench_line.json; tail -3 gpurun_out/${T}_${c}_line.err
