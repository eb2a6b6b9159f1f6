"""Would a better deal of the worlds to the physics waves pay?  (development aid; run after tools/load_study.py, which leaves
gpurun_out/load_study.npz: per step the waves' times and per-world work counters)   python tools/deal_simulation.py
A wave's time is modelled as 186 + 0.14 dd candidates + 0.84 static candidates + 0.83 dd manifolds + 5.25 x the solver
rounds of its deepest world (the fit load_study.py prints); several ways of dealing — by the previous period's counters
("pred") and by the period's own ("orac", the bound) — are scored by the mean over steps of the slowest wave."""
import numpy as np
d=np.load('gpurun_out/load_study.npz')
T=d['T'].astype(float); F=d['F'].astype(float); O=d['O'].astype(int); WR=d['WR'].astype(float)
steps,nb=T.shape; N=F.shape[1]; P=32
a_dd,a_sc,a_man,b_r=0.14,0.84,0.83,5.25
def sumcost(Fp): return a_dd*Fp[...,0]+a_sc*Fp[...,1]+a_man*Fp[...,2]
def evaluate(assign, s0):
    # assign: [N] octet of each world; evaluate modeled T over steps s0..s0+P
    mx=[];mn=[];sd=[]
    for i in range(s0,s0+P):
        sc=np.bincount(assign,weights=sumcost(F[i]),minlength=nb)
        m=np.zeros(nb); np.maximum.at(m,assign,F[i,:,3])
        t=186+sc+b_r*m
        mx.append(t.max()); mn.append(t.mean()); sd.append(t.std())
    return np.mean(mn),np.mean(sd),np.mean(mx)
def serpentine(order):
    # order: worlds sorted heaviest first; rank r -> row r//nb, col r%nb, alternate
    assign=np.zeros(N,int)
    r=np.arange(N); row=r//nb; col=r%nb
    assign[order]=np.where(row&1, nb-1-col, col)
    return assign
def greedy_rows(rows_worlds, key):
    # rows_worlds: list of arrays of nb worlds; first row assigned in order; then heaviest octet gets lightest world
    acc=np.zeros(nb); assign=np.zeros(N,int)
    for k,ws in enumerate(rows_worlds):
        ws=ws[np.argsort(key[ws])]            # ascending key
        octs=np.argsort(-acc)                 # heaviest first
        assign[ws]=octs; acc[octs]+=key[ws]
    return assign
res={}
for s0 in range(64,steps-P+1,P):
    Fp=F[s0-P:s0].sum(0)/P                 # previous period per world per step averages
    Fn=F[s0:s0+P].sum(0)/P                 # this period (oracle knowledge)
    for tag,Fx in (('pred',Fp),('orac',Fn)):
        key_cur=Fx[:,0]+Fx[:,1]
        key_sum=sumcost(Fx)
        r=Fx[:,3]
        algs={}
        algs['actual']=O[s0+P-1]
        algs['cur']=serpentine(np.argsort(-key_cur,kind='stable'))
        algs['A1 model key']=serpentine(np.argsort(-key_sum,kind='stable'))
        algs['A1b model+r key']=serpentine(np.argsort(-(key_sum+b_r*r),kind='stable'))
        top=np.argsort(-r,kind='stable')[:nb]; rest=np.setdiff1d(np.arange(N),top)
        key2=key_sum.copy(); key2[top]+=b_r*r[top]
        o_top=top[np.argsort(-key2[top],kind='stable')]; o_rest=rest[np.argsort(-key2[rest],kind='stable')]
        algs['A2 r-row serp']=serpentine(np.concatenate([o_top,o_rest]))
        rows=[o_top]+[o_rest[i*nb:(i+1)*nb] for i in range(7)]
        algs['A3 r-row greedy']=greedy_rows(rows,key2)
        o_all=np.argsort(-key_sum,kind='stable')
        algs['A4 greedy sumkey']=greedy_rows([o_all[i*nb:(i+1)*nb] for i in range(8)],key_sum)
        o_all=np.argsort(-(key_sum+b_r*r),kind='stable')
        algs['A4b greedy sum+r']=greedy_rows([o_all[i*nb:(i+1)*nb] for i in range(8)],key_sum+b_r*r)
        # grouping by r in classes of G octets, serpentine by key within class
        for G in (2000,500,250):
            if G==2000: continue
        for name,asg in algs.items():
            res.setdefault((tag,name),[]).append(evaluate(asg,s0))
for k,v in res.items():
    v=np.array(v).mean(0); print(f"{k[0]:5s} {k[1]:20s} mean {v[0]:6.1f} sd {v[1]:5.1f} max {v[2]:6.1f}")
# actual measured for comparison
sel=range(64,steps)
print('measured: mean',T[64:].mean(),'sd',T[64:].std(1).mean(),'max',T[64:].max(1).mean())
print('--- the slowest wave of a step (measured), its features vs the mean wave')
acc=[];accm=[]
for i in range(64,steps):
    o=T[i].argmax(); ws=np.nonzero(O[i]==o)[0]
    acc.append([F[i,ws,0].sum(),F[i,ws,1].sum(),F[i,ws,2].sum(),F[i,ws,3].max(),T[i,o]])
    accm.append([F[i,:,0].sum()/nb,F[i,:,1].sum()/nb,F[i,:,2].sum()/nb,WR[i].mean(),T[i].mean()])
print('slowest: ddc, sc, man, maxr, T =',np.round(np.mean(acc,0),1)); print('mean   :',np.round(np.mean(accm,0),1))
# modeled contributions
a=np.mean(acc,0); m=np.mean(accm,0)
print('model excess: ddc %.1f sc %.1f man %.1f r %.1f ; measured excess %.1f'%(0.14*(a[0]-m[0]),0.84*(a[1]-m[1]),0.83*(a[2]-m[2]),5.25*(a[3]-m[3]),a[4]-m[4]))
# distribution of per-step max r per wave
print('wave max-r percentiles (per step):',np.percentile(WR[64:],[50,90,99,99.9]), 'max',WR[64:].max())
