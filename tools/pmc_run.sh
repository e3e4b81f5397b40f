#!/bin/bash
# rocprofv3 --pmc passes (one counter group per run -- six runs --, no trace domains) of a workload; summary by kernel.
#   tools/pmc_run.sh <out-dir under gpurun_out> <kernel name filters, comma separated> -- <command...>
# Run on the GPU box from the repo root.
set -u
OUT=$1; FILTER=$2; shift 3
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$OUT
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for G in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA" \
         "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $O/pmc$i -o p -- "$@" > $O/log$i.txt 2>&1
done
cd $R
python tools/pmc_summary.py gpurun_out/$OUT ${FILTER//,/ } > gpurun_out/$OUT/summary.txt
cat gpurun_out/$OUT/summary.txt
