#!/bin/bash
# first probe: does the function form reproduce, and what do the two bisection builds do?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/fl
P=e2e-mappo-for-mt-fjsp_amd
for v in funcs funcs_vmwait funcs_poison; do
  timeout 900 python tools/first_launch/repro.py $P/libmtfjsp_$v.so 5 fused > gpurun_out/fl/$v.txt 2>&1
done
timeout 600 python tools/first_launch/repro.py $P/libmtfjsp.so 3 fused > gpurun_out/fl/textual.txt 2>&1
timeout 600 python tools/first_launch/repro.py $P/libmtfjsp_funcs.so 3 unfused > gpurun_out/fl/funcs_unfused.txt 2>&1
tail -n 3 gpurun_out/fl/*.txt
