"""Development aid: gaps between the kernels of one CAVI step, from a rocprofv3 --kernel-trace CSV of bench.py.
usage: python tools/step_gaps.py <kernel_trace.csv> [anchor-kernel-prefix]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "void agp::k_chol_dag"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows))
starts = [i for i, e in enumerate(ev) if e[2].startswith(anchor) or anchor in e[2]]
# the last 50 complete steps
starts = starts[-51:]
acc = collections.OrderedDict()
n = 0
for a, b in zip(starts[:-1], starts[1:]):
    seq = ev[a:b + 1]
    t0 = seq[0][0]
    key = tuple(e[2][:48] for e in seq[:-1])
    if n == 0:
        ref = key
    if key != ref:
        continue
    n += 1
    for j, e in enumerate(seq[:-1]):
        d = acc.setdefault(j, [e[2][:60], e[3], 0.0, 0.0, 0.0])
        d[2] += (e[0] - t0) / 1e3
        d[3] += (e[1] - t0) / 1e3
    acc.setdefault("next", ["next anchor", "", 0.0, 0.0, 0.0])[2] += (seq[-1][0] - t0) / 1e3
print(f"{n} steps averaged; times in us relative to the anchor's start")
for j, d in acc.items():
    print(f"{str(j):>4} q={d[1]:>3} start {d[2] / n:8.1f} end {d[3] / n:8.1f}  {d[0]}")
