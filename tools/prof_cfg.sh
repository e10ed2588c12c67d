# kernel statistics of a bench configuration: tools/prof_cfg.sh <config> <tag> [env...]  -> gpurun_out/stats_<tag>.csv (top rows printed)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
c=$1; T=$2
rm -rf /tmp/pc_$T
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_$T -o p -- python $R/bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline --no-elbo-tol --no-extras > /dev/null 2>&1)
f=$(find /tmp/pc_$T -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/stats_$T.csv
python - "$f" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    n=re.sub(r"void agp::|void ","",r["Name"])[:60]
    print(f'{n:60s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.1f} pct {r["Percentage"]}')
PY
