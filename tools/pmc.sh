#!/bin/bash
# developer tool: PMC counters for the step kernels (separate passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout -k 5 200 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d gpurun_out/$1 -- python3 bench.py --steps 40 --no-cpu-baseline > gpurun_out/$1.log 2>&1; }
run pmc1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
run pmc2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"
run pmc3 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES"
