#!/bin/bash
# PMC study of one conv shape: separate rocprofv3 --pmc passes (no trace domains combined), CSV per pass.
# usage: scripts/pmc_conv.sh <tag> <variant> <B> <C> <HW>
set -e
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_sum" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --output-format csv -d $R/gpurun_out/pmc_$tag/p$i -o p -- python $R/scripts/conv_micro.py "$@" > $R/gpurun_out/pmc_$tag/p$i.log 2>&1 || { mkdir -p $R/gpurun_out/pmc_$tag; echo "pass $i failed" >> $R/gpurun_out/pmc_$tag/fail.log; }
  echo "pass $i done"
done
