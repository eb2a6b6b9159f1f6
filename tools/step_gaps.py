import csv, glob, os
f=sorted(glob.glob('gpurun_out/kstats/*/*kernel_trace.csv'), key=os.path.getmtime)[-1]
rows=list(csv.DictReader(open(f)))
allk=sorted([(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0].replace('void ','')[:50]) for r in rows])
pre=[i for i,k in enumerate(allk) if 'k_physics' in k[2]]
i0=pre[len(pre)//2]; i1=pre[len(pre)//2+3]
t0=allk[i0][0]; prev=None
for s,e,k in allk[i0:i1]:
    gap = (s-prev)/1e3 if prev else 0
    print(f"{(s-t0)/1e3:8.1f} {(e-t0)/1e3:8.1f} dur {(e-s)/1e3:6.1f} gap {gap:5.1f} {k}")
    prev=e
