# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box through gpurun): bench lines, kernel statistics for c2..c5,
# PMC traffic + MFMA utilisation for c2 and c3, the hyper-on iteration.  Outputs land in gpurun_out/; copy what is to be kept into
# profiles/.   usage: bash tools/collect_profiles.sh [tag]   (default r03)
export TMPDIR=/tmp
R=$PWD
T=${1:-r04}
python bench.py > gpurun_out/${T}_c2_bench_line.json 2> gpurun_out/${T}_c2_line.err
for c in c3 c4 c5; do
  st=100; wu=20; [ $c = c5 ] && st=20 && wu=5; [ $c = c4 ] && st=60 && wu=10
  python bench.py --config $c --steps $st --warmup $wu --cpu-elbo-seconds 0 > gpurun_out/${T}_${c}_bench_line.json 2> gpurun_out/${T}_${c}_line.err
done
for c in c2 c3 c4 c5; do
  st=100; wu=20; [ $c = c5 ] && st=20 && wu=5; [ $c = c4 ] && st=60 && wu=10
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${T}_$c -o p -- python $R/bench.py --config $c --steps $st --warmup $wu --no-cpu-baseline --no-elbo-tol --no-extras > /dev/null 2>&1)
  cp gpurun_out/prof_${T}_$c/p_kernel_stats.csv gpurun_out/${T}_${c}_kernel_stats.csv
done
python tools/step_gaps.py gpurun_out/prof_${T}_c2/p_kernel_trace.csv > gpurun_out/${T}_c2_step_gaps.txt 2>&1
# the reference's default training mode: hyper-parameter step every iteration
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${T}_hyper -o p -- python $R/tools/prof_hyper.py > /dev/null 2>&1)
cp gpurun_out/prof_${T}_hyper/p_kernel_stats.csv gpurun_out/${T}_c2_hyper_kernel_stats.csv
python tools/hyper_timeline.py gpurun_out/prof_${T}_hyper/p_kernel_trace.csv > gpurun_out/${T}_c2_hyper_timeline.txt 2>&1
python tools/pmc_traffic.py collect c2 ${T}_c2 > /dev/null 2>&1; python tools/pmc_traffic.py parse c2 ${T}_c2 > gpurun_out/${T}_pmc_c2.log 2>&1
python tools/pmc_traffic.py collect c3 ${T}_c3 > /dev/null 2>&1; python tools/pmc_traffic.py parse c3 ${T}_c3 > gpurun_out/${T}_pmc_c3.log 2>&1
python tools/pmc_traffic.py mfma c2 ${T}_c2 > gpurun_out/${T}_mfma_c2.log 2>&1
for c in c2 c3 c4 c5; do echo == $c; python tools/summarize_prof.py gpurun_out/${T}_${c}_kernel_stats.csv 8; tail -c 600 gpurun_out/${T}_${c}_bench_line.json; echo; done
echo == hyper; python tools/summarize_prof.py gpurun_out/${T}_c2_hyper_kernel_stats.csv 12
cat gpurun_out/${T}_pmc_c2.log | head -30
