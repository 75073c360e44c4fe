#!/bin/bash
# Evidence for the numbers DESIGN.md cites: micro-benchmark outputs and rocprofv3 kernel stats at the larger BASELINE sizes.
#   gpurun -- 'bash tools/profile_sizes.sh r02'  ->  gpurun_out/<tag>_ubench_*.txt, <tag>_*_kernel_stats.csv, <tag>_bench_*.json
set -u
tag=${1:-prof}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
for u in bf16x6 mfma_valu; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/$u tools/ubench/$u.hip && timeout 120 /tmp/$u > gpurun_out/${tag}_ubench_$u.txt 2>&1
done
for cfg in "10x10x2 8192 J10M10E2_B8192" "20x20x4 2048 J20M20E4_B2048"; do
  set -- $cfg
  python3 bench.py --size $1 --batch $2 --steps 100 --warmup 100 --no-cpu-baseline --no-env-sweep > gpurun_out/${tag}_bench_$3.json 2> /dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$3 -- python3 bench.py --size $1 --batch $2 --steps 100 --warmup 100 --min-seconds 0.05 --no-cpu-baseline --no-env-sweep > /dev/null 2>&1
  f=$(find gpurun_out/prof_${tag}_$3 -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${tag}_$3_kernel_stats.csv
done
ls -la gpurun_out | grep ${tag}_
