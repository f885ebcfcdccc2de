import csv, glob, sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
pr=[r for r in rows if 'spgemm_topn_pruned_kernel' in r['Kernel_Name']]
t0=int(pr[-6]['Start_Timestamp']) if len(pr)>=6 else int(pr[0]['Start_Timestamp'])
for r in pr[-6:]:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    print(f"{d:10.1f} us at +{(int(r['Start_Timestamp'])-t0)/1e3:10.1f} us  {r['Kernel_Name'][:70]}  grid {r.get('Grid_Size','?')}")
