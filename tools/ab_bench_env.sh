#!/bin/bash
# A/B of the headline bench on ONE box, alternating, by an environment switch:   gpurun -- 'bash tools/ab_bench_env.sh <tag> <reps> MTFJSP_NO_ENV_PREFETCH'
# "on" = the switch unset (default path), "off" = switch=1.  Prints M env-steps/s, ms per step and the event-timed kernel times.
tag=${1:-r06}; reps=${2:-3}; var=${3:-MTFJSP_NO_ENV_PREFETCH}
mkdir -p gpurun_out
out=gpurun_out/${tag}_ab_${var}.txt
: > $out
for r in $(seq 1 $reps); do
  for v in on off; do
    if [ $v = off ]; then export $var=1; else unset $var; fi
    python bench.py --no-cpu-baseline --no-env-sweep --no-config-legs 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
kt=d.get('kernel_times_ms',{})
print('$v rep $r: %.2f M env-steps/s  %.4f ms/step  ' % (d['value']/1e6, d['ms_per_step']), {k:round(v['ms_total']/max(v['launches'],1)*1e3,2) for k,v in kt.items()})
" >> $out
  done
done
unset $var
cat $out
