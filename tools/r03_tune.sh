export TMPDIR=/tmp
for pr in l h; do
  echo "== c5 prio $pr"
  AGP_PF_PRIORITY=$pr timeout 600 python bench.py --config c5 --steps 20 --warmup 5 --no-cpu-baseline --no-elbo-tol --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
