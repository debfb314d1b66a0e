#!/bin/bash
# SQ counters of the two TFHE blind-rotation kernels (exact three-prime and fft64) at cfg5's shape, batch 1024 (tools/tfhe_fft64_lab.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_tf1 $R/gpurun_out/pmc_tf2
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_tf1 -- python3 $R/tools/tfhe_fft64_lab.py 1024 > $R/gpurun_out/pmc_tf1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_tf2 -- python3 $R/tools/tfhe_fft64_lab.py 1024 > $R/gpurun_out/pmc_tf2.log 2>&1
echo pmc_tfhe_fft64 done
