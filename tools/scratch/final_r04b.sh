#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/run_suite_and_profile.sh r04c 2>&1 | tail -40
bash tools/pmc_kernel.sh k_gin_res "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES" > gpurun_out/r04c_sq_k_gin_res.txt 2>&1
tail -20 gpurun_out/r04c_sq_k_gin_res.txt
