# Development aid: A/B of AGP_SPLIT_OVERLAP on ONE GPU with stand-in collectives (AGP_FORCE_SPLIT=1 AGP_BENCH_FAKE_ALLREDUCE_US):
# bench.py runs its timed loop once per setting of the flag in one process (collective.split_overlap_ab).
cd /root/repo
B="python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-elbo-tol --no-extras"
for us in ${@:-40 60 100}; do
  AGP_FORCE_SPLIT=1 AGP_BENCH_FAKE_ALLREDUCE_US=$us timeout 300 $B 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['collective']; ab = c['split_overlap_ab']
print('stand-in $us us: whole statistic', d['ms_per_step'], 'ms/step (', c['us_per_call'], 'us per call, host enqueue', d.get('host_enqueue_ms_per_step'), ') | column groups', ab['ms_per_step'], 'ms/step (train', ab['collective_us_per_call'], 'us )')"
done
