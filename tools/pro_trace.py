"""Development aid: prints the prologue stamps written by AGP_PRO_TRACE=<file> (agp_capi.hip), in us relative to the chain's start."""
import sys
d = {}
for l in open(sys.argv[1]):
    i, t = l.split()
    d[int(i)] = int(t)
t0 = min(d.values())
us = lambda i: (d[i] - t0) / 100.0 if i in d else float("nan")
names = ["start", "own slice", "flags", "partials", "eta2 step", "parked"]
print("chain   ", " ".join(f"{n}={us(i):7.2f}" for i, n in enumerate(names[:5])))
for R in range(4):
    for c in range(R + 1):
        b = 64 + 8 * (4 * R + c)
        if b in d:
            print(f"tile {R},{c}", " ".join(f"{n}={us(b + i):7.2f}" for i, n in enumerate(names)))
print("helper(0,0,1) start/end", us(1024), us(1025))
ks = sorted(k for k in d if 512 <= k < 600)
prev = None
for k in ks:
    print(f"factor({k - 512:2d}) done {us(k):8.2f}" + (f"  (+{us(k) - prev:6.2f})" if prev is not None else ""))
    prev = us(k)
