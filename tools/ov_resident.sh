#!/bin/bash
# AGP_SPLIT_OVERLAP on ONE GPU with a stand-in collective that must be RESIDENT to finish, like RCCL's ring kernel
# (agp_comm_standin_allreduce: W workgroups x T threads, grid barrier - copy - grid barrier, at least the requested time), on the
# communicator's own stream next to the task-graph launch whose tile workgroups wait at their arrival gates.
# usage: bash tools/ov_resident.sh            -> stdout (copy to profiles/r05_split_overlap_resident_standin.txt)
cd ${GRAFT_REPO_ROOT:-/root/repo}
B="python bench.py --warmup 30 --no-cpu-baseline --no-elbo-tol --no-extras"
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['collective']; ab = c.get('split_overlap_ab') or {}
print('  $1: whole statistic', d['ms_per_step'], 'ms/step (', c['us_per_call'], 'us per call ) | column groups', ab.get('ms_per_step'), 'ms/step ( train', ab.get('collective_us_per_call'), 'us ) | stand-in workgroups that gave up:', c.get('standin_workgroups_that_gave_up_waiting'))"; }
echo "A/B (300 steps; both settings of AGP_SPLIT_OVERLAP in one process)"
for shape in 8x256 16x512 32x512; do for us in 46 74; do
  AGP_FORCE_SPLIT=1 AGP_BENCH_FAKE_ALLREDUCE_US=$us AGP_BENCH_FAKE_RESIDENT=$shape timeout 300 $B --steps 300 2>/dev/null | tail -1 | line "resident $shape, >= $us us"
done; done
echo "soak: 30 000 steps, AGP_SPLIT_OVERLAP=1, resident stand-in 16x512 >= 46 us, host running ahead"
AGP_SPLIT_OVERLAP=1 AGP_FORCE_SPLIT=1 AGP_BENCH_FAKE_ALLREDUCE_US=46 AGP_BENCH_FAKE_RESIDENT=16x512 AGP_BENCH_NO_OVERLAP_AB=1 timeout 600 $B --steps 30000 2> gpurun_out/ov_resident_soak.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['collective']
print('  ms_per_step', d['ms_per_step'], '| train', c['us_per_call'], 'us | stand-in workgroups that gave up:', c.get('standin_workgroups_that_gave_up_waiting'), '| roofline launches per step', d['roofline']['launches_per_step'])"
grep -i "gate\|did not arrive\|fallback\|lost a tile\|status" gpurun_out/ov_resident_soak.err | head -5
echo "  (stderr of the soak: $(wc -l < gpurun_out/ov_resident_soak.err) lines; the library prints a line when an arrival gate's limit is hit (status -4) or a launch falls back)"
echo "soak with the 32x512 stand-in, 10 000 steps"
AGP_SPLIT_OVERLAP=1 AGP_FORCE_SPLIT=1 AGP_BENCH_FAKE_ALLREDUCE_US=74 AGP_BENCH_FAKE_RESIDENT=32x512 AGP_BENCH_NO_OVERLAP_AB=1 timeout 600 $B --steps 10000 2> gpurun_out/ov_resident_soak2.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['collective']
print('  ms_per_step', d['ms_per_step'], '| train', c['us_per_call'], 'us | stand-in workgroups that gave up:', c.get('standin_workgroups_that_gave_up_waiting'))"
