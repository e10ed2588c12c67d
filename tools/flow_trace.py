"""Development aid: timeline of one k_chol_flow launch (AGP_FLOW_TRACE=<file> makes the library dump wall_clock64 stamps).
usage on the GPU box:  AGP_FLOW_TRACE=/tmp/ft.txt python tools/prof_c2.py ; python tools/flow_trace.py /tmp/ft.txt"""
import sys
import numpy as np
L = open(sys.argv[1]).read().split()
nt, ne = int(L[0]), int(L[1])
t = np.array(L[2:], dtype=np.float64).reshape(nt + ne, nt + 1, 4)
t0 = t[t > 0].min()
us = lambda v: (v - t0) / 100.0 if v > 0 else float("nan")  # wall_clock64 ticks at 100 MHz
print("row  : rowdone_signal  factor_done  xready_signal   (us)")
for k in range(nt):
    print(f"{k:3d}  : {us(t[k, nt, 0]):9.1f} {us(t[k, nt, 1]):12.1f} {us(t[k, nt, 2]):12.1f}")
for row in (1, 2, nt - 1, nt, nt + ne - 1):
    print(f"row {row}: col: rowdone_seen accum_done xready_seen trsm_done")
    for j in range(min(row, nt) if row < nt else nt):
        print(f"   {j:3d}: {us(t[row, j, 0]):9.1f} {us(t[row, j, 1]):9.1f} {us(t[row, j, 2]):9.1f} {us(t[row, j, 3]):9.1f}")
