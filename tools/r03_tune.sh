export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_hyper -o p -- python $R/tools/prof_hyper.py > /dev/null 2>&1)
python tools/hyper_timeline.py gpurun_out/prof_r03_hyper/p_kernel_trace.csv | tee gpurun_out/r03_hyper_timeline_mid.txt
