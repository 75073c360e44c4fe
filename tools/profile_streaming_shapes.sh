#!/bin/bash
# Evidence for the streaming shapes: bench JSON, rocprofv3 kernel stats and HBM traffic counters (separate --pmc passes, --kernel-trace only) at
# J10M10E2 x 8192 and J20M20E4 x 2048.   gpurun -- 'bash tools/profile_streaming_shapes.sh r06'  ->  gpurun_out/<tag>_*
set -u
tag=${1:-prof}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
for cfg in "10x10x2 8192 J10M10E2_B8192" "20x20x4 2048 J20M20E4_B2048"; do
  set -- $cfg
  python3 bench.py --size $1 --batch $2 --steps 100 --warmup 100 --no-cpu-baseline --no-env-sweep > gpurun_out/${tag}_bench_$3.json 2> /dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$3 -- python3 bench.py --size $1 --batch $2 --steps 100 --warmup 100 --min-seconds 0.05 --no-cpu-baseline --no-env-sweep > /dev/null 2>&1
  f=$(find gpurun_out/prof_${tag}_$3 -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${tag}_$3_kernel_stats.csv && head -12 "$f" | cut -c1-120
  bash tools/pmc_sizes.sh ${tag} $1 $2 | grep -i "gemm_x6\|moments\|pool_gather"
  rm -rf gpurun_out/prof_${tag}_$3 gpurun_out/pmcs_${tag}_FETCH_SIZE gpurun_out/pmcs_${tag}_WRITE_SIZE      # (the raw traces: gpurun merges back at most 64 MiB)
done
