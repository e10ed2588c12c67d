cd /root/repo
for G in 2 3 6 8; do echo "groups=$G"; AGP_SPLIT_OVERLAP_GROUPS=$G bash tools/ov_ab.sh 60 100; done
echo "soak 30000 steps, overlap on, stand-in 60 us:"
AGP_SPLIT_OVERLAP=1 AGP_FORCE_SPLIT=1 AGP_BENCH_FAKE_ALLREDUCE_US=60 AGP_BENCH_NO_OVERLAP_AB=1 timeout 300 python bench.py --steps 30000 --warmup 30 --no-cpu-baseline --no-elbo-tol --no-extras 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'collective', d['collective']['us_per_call'], 'fallbacks', d.get('fallback_launches', d.get('safe_retries')))"
echo "one-rank RCCL communicator (no stand-in):"
AGP_FORCE_SPLIT=1 timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-elbo-tol --no-extras 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['collective']; print(c['issued_by'], d['ms_per_step'], c['us_per_call'], c['split_overlap_ab'])"
