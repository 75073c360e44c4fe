#!/bin/bash
# Diagnostic A/B on ONE box: the headline bench with several builds of the library, alternating, three times; HIP-event kernel times per family.
#   gpurun -- 'bash tools/ab_bench_variants.sh libmtfjsp_a.so libmtfjsp_b.so'   (paths relative to the package directory; "" = the product build)
for i in 1 2 3; do
  for lib in "$@"; do
    if [ -n "$lib" ]; then export MTFJSP_LIB=$PWD/e2e-mappo-for-mt-fjsp_amd/$lib; else unset MTFJSP_LIB; fi
    python bench.py --no-cpu-baseline --no-env-sweep --no-config-legs 2>/dev/null | tail -1 | python -c "
import sys,json,os
d=json.loads(sys.stdin.read())
k=d['kernel_times_ms']
print('%-28s'%(os.path.basename(os.environ.get('MTFJSP_LIB','product'))), 'ms/step %.5f'%d['ms_per_step'], {n:round(v['ms_total']/max(v['launches'],1)*1e3,2) for n,v in k.items() if n in('gin_resident','heads_gat3_heads','env_step')})"
  done
done
