#!/bin/bash
# developer tool: PMC counters for the step kernels (separate passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc1 -- python3 bench.py --steps 40 --no-cpu-baseline > gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/pmc2 -- python3 bench.py --steps 40 --no-cpu-baseline > gpurun_out/pmc2.log 2>&1
ls gpurun_out/pmc1/*/ gpurun_out/pmc2/*/
