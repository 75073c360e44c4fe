#!/bin/bash
# A/B on ONE box: the rollout with the environment step as the tail of the three-in-one launch (default) (MTFJSP_FUSED_ENV3=1) against the separate
# k_env_grp16 launch (default), alternating.   gpurun -- 'bash tools/ab_bench_env3.sh r06 3'
tag=${1:-r06}; reps=${2:-3}
mkdir -p gpurun_out
: > gpurun_out/${tag}_ab_env3.txt
for r in $(seq 1 $reps); do
  for v in tail separate; do
    if [ $v = tail ]; then export MTFJSP_FUSED_ENV3=1; else unset MTFJSP_FUSED_ENV3; fi
    python bench.py --no-cpu-baseline --no-env-sweep --no-config-legs 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
kt=d.get('kernel_times_ms',{})
print('$v rep $r: %.2f M env-steps/s  %.4f ms/step  ' % (d['value']/1e6, d['ms_per_step']), {k:round(v['ms_total']/max(v['launches'],1)*1e3,2) for k,v in kt.items()})
" >> gpurun_out/${tag}_ab_env3.txt
  done
done
unset MTFJSP_FUSED_ENV3
cat gpurun_out/${tag}_ab_env3.txt
