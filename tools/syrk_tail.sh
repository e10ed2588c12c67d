# Development aid: does the C3 symmetric product (528 lower tiles at m = 2048, fp32) pay for a tail?  Kernel statistics of the
# step at neighbouring tile counts: m = 1920 (465 tiles), 1984 (496), 2048 (528), 2112 (561); B = 2048 throughout.
R=${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
for m in 1920 1984 2048 2112; do
  rm -rf $R/gpurun_out/syrk_$m
  (cd /tmp && AGP_BENCH_NO_PREFETCH=${NOPF:-0} timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/syrk_$m -o p -- python $R/bench.py --config c3 --m $m --steps 60 --warmup 10 --no-cpu-baseline --no-elbo-tol --no-extras > /dev/null 2>&1)
  f=$(find $R/gpurun_out/syrk_$m -name "*kernel_stats.csv" | head -1)
  echo "m=$m tiles=$(( (m/64)*(m/64+1)/2 )): $(grep -E 'k_syrk_tn<float, 1' $f | awk -F'",' '{split($2,a,","); print "syrk avg us", a[3]/1000, "min", a[5]/1000}') | $(grep -E 'k_chol_dag<float.*, 2>' $f | awk -F'",' '{split($2,a,","); print "tile kernel avg us", a[3]/1000}')"
  rm -rf $R/gpurun_out/syrk_$m
done
