#!/bin/bash
# usage (on the GPU box): tools/pmc_run.sh <tag>   -> gpurun_out/pmc_<tag>_{a,b}/ + summary on stdout
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_${T}_a -- python3 $R/tools/one_conv.py 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM --output-format csv -d $R/gpurun_out/pmc_${T}_b -- python3 $R/tools/one_conv.py 2 > /dev/null 2>&1
python3 $R/tools/pmc_conv.py $(find $R/gpurun_out/pmc_${T}_a $R/gpurun_out/pmc_${T}_b -name "*counter_collection.csv")
