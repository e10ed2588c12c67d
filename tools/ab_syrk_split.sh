for c in c3 c4; do for s in 0 1; do
 echo -n "$c split=$s: "; AGP_SYRK_SPLIT=$s python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline --no-elbo-tol --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done; done
