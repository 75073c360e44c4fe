#!/bin/bash
# Kernel-only (rocprofv3 --kernel-trace) durations of the step kernel and of the same-footprint copy kernel, same process, per batch:
#   gpurun -- 'bash tools/footprint_kernel_only.sh r05'  ->  gpurun_out/<tag>_footprint_kernel_only.json (+ .txt)
set -u
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
for B in ${FP_SIZES:-4096 16384 65536 262144}; do
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fp_${tag}_${B} -- python3 tools/footprint_kernel_only.py --batch $B > gpurun_out/fp_${tag}_${B}.log 2>&1
done
python3 tools/footprint_reduce.py "$tag"
