# Round-4 (second half) evidence kept under profiles/ (run on the GPU box through gpurun; outputs land in gpurun_out/):
#   r04_c2_bench_line.json             the default bench line (C2) of the final build, CPU oracle run through the ELBO rules
#   r04_c2_hyper_timeline.txt / r04_c2_hyper_kernel_stats.csv   the hyper-on iteration under rocprofv3 (tools/prof_hyper.py)
#   r04_c2_kernel_stats.csv            per-kernel time of the default bench command
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/hyper_prof.sh final > /dev/null 2>&1
cp gpurun_out/hyper_tl_final.txt gpurun_out/r04_c2_hyper_timeline.txt
cp gpurun_out/hyper_stats_final.csv gpurun_out/r04_c2_hyper_kernel_stats.csv
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc2 -o p -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-elbo-tol --no-extras > /dev/null 2>&1)
cp $(find /tmp/pc2 -name "*kernel_stats.csv" | head -1) gpurun_out/r04_c2_kernel_stats.csv
python bench.py --config c3 --steps 100 --warmup 20 --cpu-elbo-seconds 0 > gpurun_out/r04b_c3_bench_line.json 2> gpurun_out/r04b_c3.err
python bench.py --config c4 --steps 60 --warmup 10 --cpu-elbo-seconds 0 > gpurun_out/r04b_c4_bench_line.json 2> gpurun_out/r04b_c4.err
python bench.py > gpurun_out/r04_c2_bench_line.json 2> gpurun_out/r04_c2_line.err
tail -c 400 gpurun_out/r04_c2_bench_line.json; tail -4 gpurun_out/r04_c2_hyper_timeline.txt
