#!/bin/bash
# SQ counter passes over the developer lab (tools/lab2), three passes of 8 SQ slots each
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq1 -- $R/tools/lab2 4096 5 > $R/gpurun_out/pmc_sq1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq2 -- $R/tools/lab2 4096 5 > $R/gpurun_out/pmc_sq2.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SMEM SQ_INSTS_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL --output-format csv -d $R/gpurun_out/pmc_sq3 -- $R/tools/lab2 4096 5 > $R/gpurun_out/pmc_sq3.log 2>&1
