#!/bin/bash
# Timing ablations of the step kernel (diagnostic builds -DMTFJSP_STAMP -DMTFJSP_STAMP_WAVES -DEG_ABL=<bits>; results are wrong by
# construction, only the stamps are read).  Builds happen in the build container (tools/ablate_env.sh build), the GPU call runs them:
#   gpurun -- 'bash tools/ablate_env.sh run r06'   ->  gpurun_out/<tag>_ablate_env.txt
# bits: 1 no global stores of the per-task part | 2 no observation-row stores | 4 no idle-term block | 8 no estimate scan / pairwise sum | 16 no LDS staging for the ELL wave
set -u
mode=${1:-run}; tag=${2:-r06}
bits=${ABL_BITS:-"0 1 2 4 8 16"}
if [ "$mode" = build ]; then
  for b in $bits; do
    python3 - "$b" <<'PY' &
import sys
sys.path.insert(0, '.')
import mtfjsp_amd
from importlib import import_module
bd = import_module('e2e-mappo-for-mt-fjsp_amd._build')
b = sys.argv[1]
print(bd.build_variant('stamp_MTFJSP_STAMP_WAVES_EG_ABL=' + b, ['-DMTFJSP_STAMP', '-DMTFJSP_STAMP_WAVES', '-DEG_ABL=' + b]))
PY
  done
  wait
  exit 0
fi
mkdir -p gpurun_out
: > gpurun_out/${tag}_ablate_env.txt
for b in $bits; do
  echo "=== EG_ABL=$b" >> gpurun_out/${tag}_ablate_env.txt
  timeout 300 python tools/stamp_env.py 6x6x2 4096 -DMTFJSP_STAMP_WAVES -DEG_ABL=$b 2>&1 | grep -a "^STAMP" | tail -19 >> gpurun_out/${tag}_ablate_env.txt
done
cat gpurun_out/${tag}_ablate_env.txt
