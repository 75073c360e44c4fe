#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r7
timeout 600 tools/ubench/mfma_raw > gpurun_out/r7/mfma_raw.txt 2>&1
grep -v "] 0  \[16-31\] 0  \[32-47\] 0  \[48-63\] 0" gpurun_out/r7/mfma_raw.txt | cut -c1-170
echo "(lines with no stale reads omitted)"
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -25
