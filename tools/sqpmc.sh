#!/bin/bash
# per-kernel SQ counters (waves, instruction mix, busy/wait cycles) of the benchmark step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/sq1 gpurun_out/sq2
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS --output-format csv -d gpurun_out/sq1 -- python3 bench.py --steps 20 --no-cpu-baseline > gpurun_out/sq1.log 2>&1 || exit 1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_IFETCH --output-format csv -d gpurun_out/sq2 -- python3 bench.py --steps 20 --no-cpu-baseline > gpurun_out/sq2.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
for d in ("sq1","sq2"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "hs::" not in k: continue
            k=k.split("(")[0].replace("void ","")
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
            cnt[(k,r["Counter_Name"])]+=1
    for k,v in agg.items():
        print(d, f"{k:28s}", " ".join(f"{c}={v[c]/cnt[(k,c)]:.0f}" for c in v))
PY
