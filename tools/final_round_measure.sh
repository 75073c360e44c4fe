#!/bin/bash
# The measurement passes behind profiles/<tag>_*: step-kernel stamps, kernel-only footprint comparison at four batches, SQ counters of the
# three rollout kernels (separate --pmc passes, --kernel-trace only).   gpurun -- 'bash tools/final_round_measure.sh r06'
set -u
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
timeout 300 python tools/stamp_env.py 6x6x2 4096 -DMTFJSP_STAMP_WAVES 2>&1 | grep -a "^STAMP" | tail -19 > gpurun_out/${tag}_stamps_k_env_grp16.txt
bash tools/footprint_kernel_only.sh ${tag} > /dev/null 2>&1
cat gpurun_out/${tag}_footprint_kernel_only.txt
bash tools/pmc_kernel.sh k_ "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES" > gpurun_out/${tag}_sq_counters_all_kernels.txt 2>&1
grep "k_gin_res\|k_headsx_gat3x_headsx\|k_env_grp16" gpurun_out/${tag}_sq_counters_all_kernels.txt | head -70
tail -3 gpurun_out/${tag}_stamps_k_env_grp16.txt
