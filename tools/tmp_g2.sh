export TMPDIR=/tmp
R=$PWD
for g in "AGP_CHOL_GROUP=4" "AGP_CHOL_GROUP=4 AGP_CHOL_LOOKAHEAD=0"; do
(cd /tmp && rm -rf /tmp/prof_g && env $g timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -o p -- python $R/bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline --no-elbo-tol --no-extras > /dev/null 2>&1 < /dev/null)
echo "== $g"; python tools/summarize_prof.py /tmp/prof_g/p_kernel_stats.csv 6 < /dev/null
done
cp /tmp/prof_g/p_kernel_trace.csv gpurun_out/g2_trace.csv
