#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
P=e2e-mappo-for-mt-fjsp_amd
timeout 300 tools/ubench/ds_write_war > gpurun_out/r4/ds_write_war.txt 2>&1
cat gpurun_out/r4/ds_write_war.txt
for v in fgat_nop1 fgat_nop2; do
  timeout 600 python tools/first_launch/warm.py $P/libmtfjsp_$v.so 4 unfused > gpurun_out/r4/warm_${v}.txt 2>&1
done
tail -n 5 gpurun_out/r4/warm_*.txt
timeout 600 python -m pytest tests/test_rollout_handoff.py -x -q -m gpu -k "timeout or lockstep or drops or rccl or exact_bn" 2>&1 | tail -5
