#!/bin/bash
# One GPU call of the step-kernel work: parity tests of the environment, in-kernel stamps, kernel-only durations beside the
# same-footprint copy.   gpurun -- 'bash tools/env_round.sh r06a [sizes...]'
set -u
tag=${1:-r06}; shift
sizes=${@:-4096 262144}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
[ -x tools/ubench/lane_moves ] && tools/ubench/lane_moves > gpurun_out/${tag}_lane_moves.txt 2>&1
timeout 1200 python -m pytest -x -q tests/test_env_hip_golden.py tests/test_env_edge_cases_gpu.py tests/test_fused_env_step_gpu.py tests/test_parallel_env_dropin_gpu.py tests/test_full_size_gpu.py -k "not encoder and not resident" > gpurun_out/${tag}_env_tests.log 2>&1
echo "env tests rc=$?" | tee -a gpurun_out/${tag}_env_tests.log
tail -3 gpurun_out/${tag}_env_tests.log
timeout 300 python tools/stamp_env.py 6x6x2 4096 -DMTFJSP_STAMP_WAVES 2>&1 | grep -a "STAMP" > gpurun_out/${tag}_stamps_env.txt
cat gpurun_out/${tag}_stamps_env.txt | tail -24
for B in $sizes; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fp_${tag}_${B} -- python3 tools/footprint_kernel_only.py --batch $B > gpurun_out/fp_${tag}_${B}.log 2>&1
done
python3 tools/footprint_reduce.py "$tag"
cat gpurun_out/${tag}_footprint_kernel_only.txt
