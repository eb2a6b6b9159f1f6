"""Print the kernel timeline of a few steps from the newest rocprofv3 kernel trace under gpurun_out/kstats."""
import csv, glob, collections, os, sys
f=sorted(glob.glob('gpurun_out/kstats/*/*kernel_trace.csv'), key=os.path.getmtime)[-1]
rows=list(csv.DictReader(open(f)))
allk=sorted([(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0].replace('void ',''), r.get('Queue_Id','?'), r.get('Stream_Id','?')) for r in rows if 'hs::' in r['Kernel_Name']])
# take one step in the middle: find k_physics occurrences
pre=[i for i,k in enumerate(allk) if 'k_physics' in k[2]]
i0=pre[len(pre)//2]; i1=pre[len(pre)//2+2] if len(pre)//2+2 < len(pre) else len(allk)
t0=allk[i0][0]
for s,e,k,q,st in allk[i0:i1][:70]:
    print(f"{(s-t0)/1e3:8.1f} {(e-t0)/1e3:8.1f} {(e-s)/1e3:6.1f} q={q} {k}")
