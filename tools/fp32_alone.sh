# Development aid: the fp32 products of the C3 step WITHOUT the look-ahead stream next to them (AGP_BENCH_NO_PREFETCH=1): kernel statistics
R=${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
rm -rf /tmp/fa; (cd /tmp && AGP_BENCH_NO_PREFETCH=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fa -o p -- python $R/bench.py --config c3 --steps 60 --warmup 10 --no-cpu-baseline --no-elbo-tol --no-extras > /dev/null 2>&1)
f=$(find /tmp/fa -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,re
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    print(f'{re.sub(r"void agp::|void ","",r["Name"])[:58]:58s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:8.1f} min {float(r["MinNs"])/1e3:8.1f}')
PY
