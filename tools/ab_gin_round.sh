#!/bin/bash
# One GPU call of the k_gin_res work: encoder parity tests, A/B of the headline bench (product build against a variant, alternating, one box), phase stamps of both.
#   gpurun -- 'bash tools/ab_gin_round.sh r06p libmtfjsp_ab_wpre0.so "libmtfjsp_grstamp0.so libmtfjsp_grstamp0_GR_WPRE=0.so"'
set -u
tag=$1; base=$2; stamps=${3:-}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
timeout 1500 python -m pytest -x -q tests/test_encoder_hip.py tests/test_resident_statistics_gpu.py tests/test_full_size_gpu.py tests/test_encoder_sizes_gpu.py > gpurun_out/${tag}_enc_tests.log 2>&1
echo "encoder tests rc=$?" | tee -a gpurun_out/${tag}_enc_tests.log
tail -3 gpurun_out/${tag}_enc_tests.log
bash tools/ab_bench_variants.sh "" $base 2>&1 | tee gpurun_out/${tag}_ab.txt
for s in $stamps; do
  echo "== $s" | tee -a gpurun_out/${tag}_stamps.txt
  MTFJSP_LIB=$PWD/e2e-mappo-for-mt-fjsp_amd/$s MTFJSP_STAMP_PRINT=1 timeout 300 python bench.py --steps 360 --warmup 360 --min-seconds 0.5 --no-cpu-baseline --no-env-sweep --no-config-legs 2>&1 | grep -a "GR_STAMP" | tail -2 | tee -a gpurun_out/${tag}_stamps.txt
done
