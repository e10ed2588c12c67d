"""Development aid: kernel timeline of ONE hyper-on training iteration (CAVI step + hyper step + K refresh) from a rocprofv3
--kernel-trace CSV of tools/prof_hyper.py: iterations are cut at the step's first kernel matrix launch, the last complete ones averaged.
usage: python tools/hyper_timeline.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"void agp::|void ", "", r["Kernel_Name"])[:70]) for r in rows)
# anchor: the Knm launch of the step (k_kernelmatrix_mma<double, 0, 2>) -- one per iteration in hyper-on mode
anchors = [i for i, e in enumerate(ev) if e[2].startswith("k_kernelmatrix_mma<double, 0, 2>")]
anchors = anchors[-22:]
seqs = [ev[a:b] for a, b in zip(anchors[:-1], anchors[1:])]
ref = tuple(e[2][:40] for e in seqs[-1])
seqs = [s for s in seqs if tuple(e[2][:40] for e in s) == ref]
n = len(seqs)
print(f"{n} iterations averaged, {len(ref)} kernels each")
tot_busy = 0.0
prev_end = 0.0
for j in range(len(ref)):
    st = sum(s[j][0] - s[0][0] for s in seqs) / n / 1e3
    en = sum(s[j][1] - s[0][0] for s in seqs) / n / 1e3
    tot_busy += en - st
    print(f"{j:3d} start {st:8.1f} dur {en - st:7.1f} gap {st - prev_end:6.1f}  {seqs[-1][j][2]}")
    prev_end = en
per = sum(b[0][0] - a[0][0] for a, b in zip(seqs[:-1], seqs[1:])) / max(n - 1, 1) / 1e3
print(f"iteration period {per:.1f} us, kernel time {tot_busy:.1f} us")
