export TMPDIR=/tmp
B="python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-elbo-tol --no-extras"
for mg in 1 0; do for us in 20 40 80; do
echo "== merged=$mg split fake $us"; AGP_SPLIT_MERGED=$mg AGP_FORCE_SPLIT=1 AGP_BENCH_FAKE_ALLREDUCE_US=$us $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['collective']['us_per_call'], d.get('step_counters'))"
done; done
timeout 1500 python -m pytest tests/test_gpu_comm.py tests/test_gpu_round3.py tests/test_gpu_round2.py -m gpu -x -q 2>&1 | grep -v RCCL | tail -5
