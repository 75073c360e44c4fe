#!/bin/bash
# A/B of the two `configs` legs between the product library and a build variant (MTFJSP_LIB), alternating, one box.
#   gpurun -- 'bash tools/ab_config_legs_lib.sh r06 libmtfjsp_ab_x.so 2'
tag=$1; lib=$2; reps=${3:-2}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out; : > gpurun_out/${tag}_ab_config_legs_lib.txt
for i in $(seq 1 $reps); do
  echo "== product" >> gpurun_out/${tag}_ab_config_legs_lib.txt; python tools/bench_config_leg.py 2>/dev/null | grep '^{' >> gpurun_out/${tag}_ab_config_legs_lib.txt
  echo "== $lib" >> gpurun_out/${tag}_ab_config_legs_lib.txt; MTFJSP_LIB=$PWD/e2e-mappo-for-mt-fjsp_amd/$lib python tools/bench_config_leg.py 2>/dev/null | grep '^{' >> gpurun_out/${tag}_ab_config_legs_lib.txt
done
cat gpurun_out/${tag}_ab_config_legs_lib.txt
