export TMPDIR=/tmp
B="python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-elbo-tol --no-extras"
echo "== fused"; $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
for us in 0 20 40 80; do
echo "== split fake $us"; AGP_FORCE_SPLIT=1 AGP_BENCH_FAKE_ALLREDUCE_US=$us $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('collective'))"
done
