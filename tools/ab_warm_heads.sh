for i in 1 2 3; do
  for v in 1 0; do
    if [ $v = 1 ]; then export MTFJSP_NO_WARM_HEADS=1; else unset MTFJSP_NO_WARM_HEADS; fi
    python bench.py --no-cpu-baseline --no-env-sweep --no-config-legs 2>/dev/null | tail -1 | python -c "
import sys,json,os
d=json.loads(sys.stdin.read())
k=d['kernel_times_ms']
print('nowarm' if os.environ.get('MTFJSP_NO_WARM_HEADS') else 'warm  ', 'ms/step %.5f'%d['ms_per_step'], {n:round(v['ms_total']/max(v['launches'],1)*1e3,2) for n,v in k.items() if n in('gin_resident','heads_gat3_heads','env_step')})"
  done
done
