#!/bin/bash
# instruction-cache behaviour of the step kernels (SQC counters), sequential launches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ic1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/ic1 -- python3 bench.py --steps 20 --warmup 100 --no-cpu-baseline > gpurun_out/ic1.log 2>&1 || { tail -5 gpurun_out/ic1.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob("gpurun_out/ic1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "hs::" not in k: continue
        k=k.split("(")[0].replace("void ","")
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k,v in agg.items():
    print(f"{k:28s}", " ".join(f"{c}={v[c]/cnt[(k,c)]:.0f}" for c in v))
PY
