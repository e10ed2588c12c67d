#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_s3; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -m gpu 2>&1 | tail -25 ) > $O/tests_r6.txt
( timeout 900 python bench.py --no-cpu-baseline 2>$O/bench_c2.err | tail -1 ) > $O/bench_c2.json
( AGP_BENCH_ELBO_INLINE=1 timeout 900 python bench.py --no-cpu-baseline --no-extras 2>$O/bench_c2_inline.err | tail -1 ) > $O/bench_c2_inline.json
( timeout 1200 python tools/mfma_ceiling.py $O/mfma_ceiling.txt > /dev/null 2>$O/mfma_ceiling.err )
# counters of the streaming predictor
cat > /tmp/pred.py <<'PY'
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
import __graft_entry__ as g; g.build()
import agp_amd as AGP
from agp_amd import capi
rng = np.random.default_rng(0)
N, D, m, B = 1000000, 32, 1024, 1024
X = torch.rand(N, D, dtype=torch.float64, device="cuda")
y = torch.sign(torch.randn(N, dtype=torch.float64, device="cuda"))
Z = X[:m].cpu().numpy()
model = AGP.SVGP(AGP.with_lengthscale(AGP.SqExponentialKernel(), 1.4), AGP.LogisticLikelihood(), AGP.AnalyticSVI(B), Z, optimiser=False)
Xh = X.cpu().numpy(); yh = y.cpu().numpy()
AGP.train_(model, Xh, yh, 3)
L = capi.lib(); h = model._h
Xd, yd, _ = model._data
out = torch.empty(1, N, dtype=torch.float64, device="cuda")
for _ in range(5):
    model._chk(L.agp_svgp_predict_f(h, C.c_void_p(Xd.data_ptr()), Xd.stride(0), N, C.c_void_p(out.data_ptr()), None))
torch.cuda.synchronize()
PY
cd /tmp
( rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/pred_stats -o pred -- python /tmp/pred.py > /dev/null 2>&1 )
for pm in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  n=$(echo $pm | tr ' ' '_')
  ( rocprofv3 --pmc $pm -d $GRAFT_REPO_ROOT/$O/pred_pmc_$n -o pred -- python /tmp/pred.py > /dev/null 2>&1 )
done
cd $GRAFT_REPO_ROOT
find $O -name "*.csv" | head -30
