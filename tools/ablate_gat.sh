#!/bin/bash
# Diagnostic: in-kernel stamps of the three-in-one heads launch with parts of the GAT tile loop removed (wrong results; timing only).
#   gpurun -- 'bash tools/ablate_gat.sh "0 1 8 31"'   -> gpurun_out/ablate_gat.txt
mkdir -p gpurun_out
for abl in ${1:-0 1 2 4 8 16 31}; do
  echo "== GAT_ABL=$abl" >> gpurun_out/ablate_gat.txt
  python tools/stamp_fused3.py -DGAT_ABL=$abl 2>&1 | grep "STAMP3" | head -15 >> gpurun_out/ablate_gat.txt
done
cat gpurun_out/ablate_gat.txt
