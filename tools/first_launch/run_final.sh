#!/bin/bash
# the function-form build (-DMTFJSP_BODY_FUNCS=3) after the fix: warm decisions, then 20 cold starts fused + 6 unfused, and the textual product
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04fl
P=e2e-mappo-for-mt-fjsp_amd
timeout 600 python tools/first_launch/warm.py $P/libmtfjsp_funcs.so 8 fused 2>&1 | grep -v amdgpu.ids > gpurun_out/r04fl/warm_funcs_fused.txt
timeout 600 python tools/first_launch/warm.py $P/libmtfjsp_funcs.so 8 unfused 2>&1 | grep -v amdgpu.ids > gpurun_out/r04fl/warm_funcs_unfused.txt
timeout 600 python tools/first_launch/warm.py $P/libmtfjsp.so 8 fused 2>&1 | grep -v amdgpu.ids > gpurun_out/r04fl/warm_product_fused.txt
tail -n 2 gpurun_out/r04fl/warm_*.txt
timeout 1500 python tools/first_launch/repro.py $P/libmtfjsp_funcs.so 20 fused 2>&1 | grep -v amdgpu.ids > gpurun_out/r04fl/cold_funcs_fused.txt
timeout 600 python tools/first_launch/repro.py $P/libmtfjsp_funcs.so 6 unfused 2>&1 | grep -v amdgpu.ids > gpurun_out/r04fl/cold_funcs_unfused.txt
tail -n 1 gpurun_out/r04fl/cold_*.txt
tools/ubench/valu_after_mfma > gpurun_out/r04fl/valu_after_mfma.txt 2>&1
