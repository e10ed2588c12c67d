for spec in "c2 1536 1536" "c2 2048 2048" "c2 1024 2048" "c2 1024 4096" "c3 1024 2048" "c3 1536 1536" "c3 2048 4096"; do
  set -- $spec
  for sp in 0 1; do
    line=$(AGP_CHAIN_SPLIT=$sp AGP_STEP_PROLOGUE=${PRO:-1} timeout 200 python bench.py --config $1 --m $2 --batch $3 --steps 60 --warmup 10 --no-cpu-baseline --no-elbo-tol --no-extras 2>/dev/null | tail -1)
    echo "$spec split=$sp $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"].get("avg_launch_us"))')"
  done
done
