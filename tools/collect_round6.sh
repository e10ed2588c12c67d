#!/bin/bash
# Round-6 evidence kept under profiles/ (run on the GPU box through gpurun; outputs land in gpurun_out/, copy r06_* to profiles/):
#   r06_c{2..5}_bench_line.json     the JSON line of bench.py --config cN (no profiler attached), FINAL build
#   r06_c{2..5}_kernel_stats.csv    rocprofv3 --kernel-trace --stats of the same commands
#   r06_c2_step_gaps.txt, r06_c4_step_trace.txt, r06_c2_hyper_timeline.txt, r06_c2_hyper_kernel_stats.csv
#   r06_c{2..5}_pmc_hbm_bytes.json, r06_hyper_pmc_hbm_bytes.json, r06_c2_pmc_mfma_util.json, r06_c4_pmc_mfma_util.json, r06_hyper_pmc_mfma_util.json
#   r06_soaks.txt                   determinism soaks of the final build
# usage: bash tools/collect_round6.sh [pmc|lines|stats|soaks ...]   (default: everything; PMC passes FIRST so that the bench lines
#        printed afterwards carry their traffic figures)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out profiles
T=r06
WHAT="${@:-pmc lines stats predict soaks}"
steps() { st=100; wu=20; [ $1 = c5 ] && st=20 && wu=5; [ $1 = c4 ] && st=60 && wu=10; echo "--steps $st --warmup $wu"; }
for w in $WHAT; do case $w in
pmc)
  for c in c2 c3 c4 c5 hyper; do
    python tools/pmc_traffic.py collect $c ${T}_$c > /dev/null 2>&1
    python tools/pmc_traffic.py parse $c ${T}_$c > gpurun_out/${T}_pmc_$c.log 2>&1
    cp gpurun_out/pmc_${T}_$c/${T}_${c}_pmc_hbm_bytes.json gpurun_out/ 2>/dev/null && cp gpurun_out/${T}_${c}_pmc_hbm_bytes.json profiles/
  done
  for c in c2 c4 hyper; do
    python tools/pmc_traffic.py mfma $c ${T}_$c > gpurun_out/${T}_mfma_$c.log 2>&1
    cp gpurun_out/pmc_${T}_$c/${T}_${c}_pmc_mfma_util.json gpurun_out/ 2>/dev/null
  done ;;
lines)
  python bench.py > gpurun_out/${T}_c2_bench_line.json 2> gpurun_out/${T}_c2_line.err
  for c in c3 c4 c5; do python bench.py --config $c $(steps $c) --cpu-elbo-seconds 0 > gpurun_out/${T}_${c}_bench_line.json 2> gpurun_out/${T}_${c}_line.err; done ;;
stats)
  for c in c2 c3 c4 c5; do
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${T}_$c -o p -- python $R/bench.py --config $c $(steps $c) --no-cpu-baseline --no-elbo-tol --no-extras > /dev/null 2>&1)
    cp $(find gpurun_out/prof_${T}_$c -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_${c}_kernel_stats.csv
  done
  python tools/step_gaps.py $(find gpurun_out/prof_${T}_c2 -name "*kernel_trace.csv" | head -1) > gpurun_out/${T}_c2_step_gaps.txt 2>&1
  # C4: the step's launches on all four streams relative to the tile kernel's start (where the 1.08 ms go)
  python tools/step_gaps.py $(find gpurun_out/prof_${T}_c4 -name "*kernel_trace.csv" | head -1) "k_chol_dag<double, true, true, false, true, false, 2>" > gpurun_out/${T}_c4_step_trace.txt 2>&1
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${T}_hyper -o p -- python $R/tools/prof_hyper.py > /dev/null 2>&1)
  cp $(find gpurun_out/prof_${T}_hyper -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_c2_hyper_kernel_stats.csv
  python tools/hyper_timeline.py $(find gpurun_out/prof_${T}_hyper -name "*kernel_trace.csv" | head -1) > gpurun_out/${T}_c2_hyper_timeline.txt 2>&1 ;;
predict)
  # the streaming predictor alone: kernel statistics of five passes over N = 1e6 at the C2 shape (round 6: predict_roofline of the bench line)
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${T}_predict -o p -- python $R/tools/prof_predict.py > /dev/null 2>&1)
  cp $(find gpurun_out/prof_${T}_predict -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_predict_kernel_stats.csv
  python tools/bench_diag.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_tile_factorisation_decomposition.txt ;;
soaks)
  { echo "# final build of round 6";
    echo "## C2 shape fp64, 2 x 60000 steps (merged launch, look-ahead on)"; timeout 900 python tools/soak_determinism.py 60000;
    echo "## C3 shape fp32 m = B = 2048, 2 x 20000 steps (split launch)"; timeout 900 python tools/soak_determinism.py 20000 2048 f32;
    echo "## fp32 m = 1024, B = 2048: split launch WITH the prologue (default again since round 6), 2 x 20000 steps"; timeout 900 python tools/soak_determinism.py 20000 1024 f32 cavi 2048;
    echo "## 8 latents (C4 shape: split launch of 8 chains + tiles, two look-ahead streams, lane-parallel k_lsm_fused), 2 x 5000 steps"; timeout 900 python tools/soak_multilatent.py 8 5000 2;
    echo "## hyper-on iteration, 2 x 2000"; timeout 900 python tools/soak_hyper.py 2000; } 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_soaks.txt ;;
esac; done
rm -rf gpurun_out/prof_${T}_* gpurun_out/pmc_${T}_*/bench_* gpurun_out/pmc_${T}_*/cal_* 2>/dev/null
ls gpurun_out | grep ${T}_ | head -40
