import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
ks=[i for i,r in enumerate(rows) if 'k_rowstats_local' in r['Kernel_Name']]
a,b=ks[-3],ks[-2]
t0=int(rows[a]['Start_Timestamp'])
for r in rows[a:b+1]:
    n=r['Kernel_Name'].replace('void agp::','')[:24]
    print(f"{n:26s} q{r.get('Queue_Id','?'):>3s} start {(int(r['Start_Timestamp'])-t0)/1e3:8.1f} dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f}")
