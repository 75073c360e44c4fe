#!/bin/bash
# A/B on ONE box, alternating: the per-episode work of round 6 (one reset launch per episode; the post-terminal forward pair without its
# scorers) against round 5's (three launches; full heads).   gpurun -- 'bash tools/ab_bench_episode.sh r06 3'
tag=${1:-r06}; reps=${2:-3}
mkdir -p gpurun_out
: > gpurun_out/${tag}_ab_episode.txt
for r in $(seq 1 $reps); do
  for v in new old; do
    if [ $v = old ]; then export MTFJSP_NO_RESET_EPISODE=1 MTFJSP_NO_VALUES_ONLY=1; else unset MTFJSP_NO_RESET_EPISODE MTFJSP_NO_VALUES_ONLY; fi
    python bench.py --no-cpu-baseline --no-env-sweep --no-config-legs 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
kt=d.get('kernel_times_ms',{})
print('$v rep $r: %.2f M env-steps/s  %.4f ms/step  ' % (d['value']/1e6, d['ms_per_step']), {k:(round(v['ms_total']/max(v['launches'],1)*1e3,2), v['launches']) for k,v in kt.items()})
" >> gpurun_out/${tag}_ab_episode.txt
  done
done
unset MTFJSP_NO_RESET_EPISODE MTFJSP_NO_VALUES_ONLY
cat gpurun_out/${tag}_ab_episode.txt
