#!/bin/bash
# Diagnostic A/B on ONE box: the stamps of the three-in-one heads launch for several sets of extra compile flags, alternating, twice.
#   gpurun -- 'bash tools/ab_fused3.sh "" "-DHX_C_UNROLL=6"'   -> gpurun_out/ab_fused3.txt
mkdir -p gpurun_out
for rep in 1 2; do
  for flags in "$@"; do
    echo "== flags: [$flags] (run $rep)" >> gpurun_out/ab_fused3.txt
    python tools/stamp_fused3.py $flags 2>&1 | grep "STAMP3" | head -24 >> gpurun_out/ab_fused3.txt
  done
done
cat gpurun_out/ab_fused3.txt
