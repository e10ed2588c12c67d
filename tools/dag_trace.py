"""Development aid: timeline of one k_chol_dag launch (AGP_DAG_TRACE=<file> makes the library dump wall_clock64 stamps, 4 per
tile workgroup: start, pending updates done, [diag: factored | other: X_c seen], published).
usage on the GPU box:  AGP_CHOL_DAG=1 AGP_DAG_TRACE=/tmp/dt.txt python tools/prof_c2.py ; python tools/dag_trace.py /tmp/dt.txt"""
import sys

import numpy as np

L = open(sys.argv[1]).read().split()
nt, ne = int(L[0]), int(L[1])
t = np.array(L[2:], dtype=np.float64).reshape(-1, 8)
t0 = t[:, :7][t[:, :7] > 0].min()
us = lambda v: (v - t0) / 100.0 if v > 0 else float("nan")  # noqa: E731  wall_clock64 ticks at 100 MHz


def wg(c, b):
    return sum(nt - cc + ne for cc in range(c)) + b


print("col : diag start | pending done | factored | X published || tile (c+1,c): pending done | X seen | L published   (us)")
prev = None
for c in range(nt):
    d = t[wg(c, 0)]
    line = f"{c:3d} : {us(d[0]):8.1f} {us(d[1]):8.1f} {us(d[2]):8.1f} {us(d[3]):8.1f}"
    if c + 1 < nt:
        n = t[wg(c, 1)]
        line += f" || {us(n[1]):8.1f} {us(n[2]):8.1f} {us(n[3]):8.1f}"
    if prev is not None:
        line += f"   [column period {us(d[2]) - prev:5.1f}]"
    prev = us(d[2])
    print(line)
print("col : chain detail (us): X published -> (c+1,c) saw X +d | X in LDS +d | product+stores issued +d | L published +d || diag(c+1): "
      "flag seen +d | L in LDS +d | product done +d | factored +d")
for c in range(nt - 1):
    d, n, e = t[wg(c, 0)], t[wg(c, 1)], t[wg(c + 1, 0)]
    seq = [d[3], n[2], n[6], n[7], n[3], e[4], e[5], e[1], e[2]]
    print(f"{c:3d} : " + " ".join(f"{(b - a) / 100.0:5.2f}" for a, b in zip(seq[:-1], seq[1:])))
print("fused chain (us): col | T, D in LDS (prefetched or fetched) +d | L = T X' MFMAs, X published +d | L stored +d | S = D - L L' in LDS +d | factored +d"
      " || prefetch hit, D parked this long before the factorisation of the column ended (us)")
for c in range(nt - 1):
    d, e, f1 = t[wg(c, 0)], t[wg(c + 1, 0)], t[wg(c, 1)]
    seq = [d[2], e[4], d[3], e[5], e[1], e[2]]
    if min(seq) > 0:
        print(f"{c:3d} : " + " ".join(f"{(b - a) / 100.0:5.2f}" for a, b in zip(seq[:-1], seq[1:])) +
              f" || {int(e[7])} {(d[2] - e[6]) / 100.0:6.1f}")
last = t[:, 3].max()
print(f'chain: first factorisation done {us(t[wg(0, 0)][2]):.1f}, last {us(t[wg(nt - 1, 0)][2]):.1f} -> {(us(t[wg(nt - 1, 0)][2]) - us(t[wg(0, 0)][2])) / (nt - 1):.2f} us per column')
print(f"last tile published at {us(last):.1f} us ; first start {us(t[:, 0].min()):.1f} ; latest start {us(t[:, 0].max()):.1f}")
