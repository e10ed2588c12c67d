# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box through gpurun): bench lines, kernel statistics for c2..c5,
# PMC traffic for c2 and c3.  Outputs land in gpurun_out/; copy what is to be kept into profiles/.
export TMPDIR=/tmp
R=$PWD
for c in c2 c3 c4 c5; do
  st=100; wu=20; [ $c = c5 ] && st=20 && wu=5; [ $c = c4 ] && st=60 && wu=10
  python bench.py --config $c --steps $st --warmup $wu > gpurun_out/r02_${c}_line.json 2> gpurun_out/r02_${c}_line.err
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02_$c -o p -- python $R/bench.py --config $c --steps $st --warmup $wu --no-cpu-baseline --no-elbo-tol --no-extras > /dev/null 2>&1)
  cp gpurun_out/prof_r02_$c/p_kernel_stats.csv gpurun_out/r02_${c}_kernel_stats.csv
done
python tools/pmc_traffic.py collect c2 r02_c2 > /dev/null 2>&1; python tools/pmc_traffic.py parse c2 r02_c2 > gpurun_out/r02_pmc_c2.log 2>&1
python tools/pmc_traffic.py collect c3 r02_c3 > /dev/null 2>&1; python tools/pmc_traffic.py parse c3 r02_c3 > gpurun_out/r02_pmc_c3.log 2>&1
for c in c2 c3 c4 c5; do echo == $c; python tools/summarize_prof.py gpurun_out/r02_${c}_kernel_stats.csv 8; tail -c 700 gpurun_out/r02_${c}_line.json; echo; done
