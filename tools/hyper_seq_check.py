"""Development aid: kernel counts per hyper-on iteration in a rocprofv3 kernel trace (tools/prof_hyper.py), and where iterations differ."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"void agp::|void ", "", r["Kernel_Name"])[:50]) for r in rows)
anchors = [i for i, e in enumerate(ev) if e[2].startswith("k_kernelmatrix_mma<double, 0, 2>")]
seqs = [ev[a:b] for a, b in zip(anchors[:-1], anchors[1:])]
cnt = collections.Counter(len(s) for s in seqs)
print("kernels per iteration:", dict(cnt))
per = [(b[0][0] - a[0][0]) / 1e3 for a, b in zip(seqs[:-1], seqs[1:])]
print("periods (us), last 12:", [round(p, 1) for p in per[-12:]])
long_ = [s for s in seqs[-20:] if len(s) != min(cnt)]
if long_:
    base = [e[2] for e in min(seqs[-20:], key=len)]
    for e in long_[-1]:
        if e[2] not in base:
            print("extra:", e[2], (e[1] - e[0]) / 1e3)
