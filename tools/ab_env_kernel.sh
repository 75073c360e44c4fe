#!/bin/bash
# A/B of step-kernel builds on ONE box: tools/footprint_kernel_only.py (step kernel between event records / in the rollout's pattern,
# beside the same-footprint copy) under rocprofv3 --kernel-trace for each library in turn, repeated (devices of the pool differ by
# 5-10 %: only same-box numbers compare).
#   gpurun -- 'bash tools/ab_env_kernel.sh r06 3 4096 libmtfjsp_ab_r05.so libmtfjsp.so'  ->  gpurun_out/ab_<tag>_summary.txt
set -u
tag=$1; reps=$2; B=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
: > gpurun_out/ab_${tag}_summary.txt
for r in $(seq 1 $reps); do
  for lib in "$@"; do
    name=$(basename $lib .so)
    t=${tag}-${name}-${r}
    MTFJSP_LIB=$PWD/e2e-mappo-for-mt-fjsp_amd/$lib timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fp_${t}_${B} -- python3 tools/footprint_kernel_only.py --batch $B > gpurun_out/fp_${t}_${B}.log 2>&1
    python3 tools/footprint_reduce.py "$t" > /dev/null
    echo "$name rep $r: $(grep '^B=' gpurun_out/${t}_footprint_kernel_only.txt)" >> gpurun_out/ab_${tag}_summary.txt
  done
done
sort gpurun_out/ab_${tag}_summary.txt | cut -c1-330
