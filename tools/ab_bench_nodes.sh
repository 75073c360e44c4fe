#!/bin/bash
# A/B on ONE box, alternating: the three-in-one launch with the GAT part's node rows kept in LDS (round 6, default) against the round-5
# form (MTFJSP_FUSED3_NODES_HBM=1: rows written to memory and read back).   gpurun -- 'bash tools/ab_bench_nodes.sh r06 3'
tag=${1:-r06}; reps=${2:-3}
mkdir -p gpurun_out
: > gpurun_out/${tag}_ab_nodes.txt
for r in $(seq 1 $reps); do
  for v in lds hbm; do
    if [ $v = hbm ]; then export MTFJSP_FUSED3_NODES_HBM=1; else unset MTFJSP_FUSED3_NODES_HBM; fi
    python bench.py --no-cpu-baseline --no-env-sweep --no-config-legs 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
kt=d.get('kernel_times_ms',{})
print('$v rep $r: %.2f M env-steps/s  %.4f ms/step  ' % (d['value']/1e6, d['ms_per_step']), {k:round(v['ms_total']/max(v['launches'],1)*1e3,2) for k,v in kt.items()})
" >> gpurun_out/${tag}_ab_nodes.txt
  done
done
unset MTFJSP_FUSED3_NODES_HBM
cat gpurun_out/${tag}_ab_nodes.txt
