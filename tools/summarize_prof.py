import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", round(tot / 1e6, 3))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    n = re.sub(r'void agp::', '', r['Name'])[:60]
    print(f"{n:62s} calls {int(r['Calls']):6d} avg_us {float(r['AverageNs'])/1e3:9.2f} tot_ms {float(r['TotalDurationNs'])/1e6:8.2f} {float(r['Percentage']):6.2f}%")
