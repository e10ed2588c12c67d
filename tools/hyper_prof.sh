# kernel timeline of the hyper-on iteration (tools/prof_hyper.py under rocprofv3) -> gpurun_out/hyper_tl_$1.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-x}
rm -rf /tmp/ph_$T
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ph_$T -o p -- python $R/tools/prof_hyper.py > /dev/null 2>&1)
f=$(find /tmp/ph_$T -name "*kernel_trace.csv" | head -1)
python $R/tools/hyper_timeline.py $f > $R/gpurun_out/hyper_tl_$T.txt 2>&1
python $R/tools/hyper_seq_check.py $f >> $R/gpurun_out/hyper_tl_$T.txt 2>&1
cp $(find /tmp/ph_$T -name "*kernel_stats.csv" | head -1) $R/gpurun_out/hyper_stats_$T.csv
cat $R/gpurun_out/hyper_tl_$T.txt
