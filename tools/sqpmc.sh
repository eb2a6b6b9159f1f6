#!/bin/bash
# per-kernel SQ counters (waves, instruction mix, busy/wait cycles, instruction fetch, LDS) of the benchmark step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/sq1 gpurun_out/sq2 gpurun_out/sq3 gpurun_out/sq4
run() { d=$1; shift; timeout -k 5 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/$d -- python3 bench.py --steps 24 --warmup 5 --no-cpu-baseline > gpurun_out/$d.log 2>&1 || { tail -5 gpurun_out/$d.log; exit 1; }; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM
run sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_IFETCH SQ_WAIT_ANY
run sq3 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
run sq4 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM
python3 - <<'PY'
import csv, glob, collections
for d in ("sq1","sq2","sq3","sq4"):
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
