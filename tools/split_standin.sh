# Development aid: the batch-parallel (phase-split) CAVI step on ONE GPU with a stand-in kernel in the all-reduce's place
# (AGP_FORCE_SPLIT=1 + AGP_BENCH_FAKE_ALLREDUCE_US), with and without the round-3 scheduling (AGP_SPLIT_MERGED); profiles/r03_split_standin.txt
B="python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-elbo-tol --no-extras"
echo "fused one-GPU step:"; $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ms_per_step', d['ms_per_step'])"
for mg in 1 0; do for us in 20 40 80; do
  AGP_SPLIT_MERGED=$mg AGP_FORCE_SPLIT=1 AGP_BENCH_FAKE_ALLREDUCE_US=$us $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('AGP_SPLIT_MERGED=$mg stand-in requested $us us: measured', d['collective']['us_per_call'], 'us per call -> ms_per_step', d['ms_per_step'])"
done; done
