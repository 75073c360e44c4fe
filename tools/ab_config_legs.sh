#!/bin/bash
# A/B of the two `configs` legs (J10M10E2 x 8192, J20M20E4 x 2048) by an environment switch, alternating, one box.
#   gpurun -- 'bash tools/ab_config_legs.sh r06 MTFJSP_FUSE_GIN0=0 2'
tag=$1; sw=$2; reps=${3:-2}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out; : > gpurun_out/${tag}_ab_config_legs.txt
for i in $(seq 1 $reps); do
  echo "== product" >> gpurun_out/${tag}_ab_config_legs.txt; python tools/bench_config_leg.py 2>/dev/null | grep '^{' >> gpurun_out/${tag}_ab_config_legs.txt
  echo "== $sw" >> gpurun_out/${tag}_ab_config_legs.txt; env $sw python tools/bench_config_leg.py 2>/dev/null | grep '^{' >> gpurun_out/${tag}_ab_config_legs.txt
done
cat gpurun_out/${tag}_ab_config_legs.txt
