export TMPDIR=/tmp
R=$PWD
B="python $R/bench.py --config c4 --steps 60 --warmup 10 --no-cpu-baseline --no-elbo-tol --no-extras"
for f in 1 0; do echo "== c4 lsm_fused=$f"; AGP_LSM_FUSED=$f timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['launches_per_step'])"; done
timeout 900 python tools/soak_multilatent.py 8 3000 2 2>&1 | tail -4
timeout 900 python tools/soak_multilatent.py 5 2000 2 2>&1 | tail -3
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
