#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3
P=e2e-mappo-for-mt-fjsp_amd
for v in fgat_wz fgat_pad; do
  timeout 600 python tools/first_launch/warm.py $P/libmtfjsp_$v.so 4 unfused > gpurun_out/r3/warm_${v}.txt 2>&1
done
tail -n 5 gpurun_out/r3/warm_*.txt
for pm in 0 1; do
MTFJSP_GIN_RES_POLL=$pm MTFJSP_LIB=$P/libmtfjsp_grstamp0.so MTFJSP_STAMP_PRINT=1 timeout 300 python bench.py --steps 72 --warmup 36 --min-seconds 0.01 --no-cpu-baseline --no-env-sweep --no-config-legs 2>&1 | grep GR_STAMP > gpurun_out/r3/stamps_poll$pm.txt
done
cat gpurun_out/r3/stamps_poll*.txt
