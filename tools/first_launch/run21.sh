#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r21
P=e2e-mappo-for-mt-fjsp_amd
timeout 900 python -m pytest tests/test_encoder_hip.py tests/test_full_size_gpu.py tests/test_first_launch_gpu.py -x -q -m gpu -k "encoder or resident or actor or environment_step" 2>&1 | tail -4
for i in 1 2; do
timeout 300 python bench.py --no-config-legs --no-env-sweep --no-cpu-baseline > gpurun_out/r21/bench$i.json 2> gpurun_out/r21/bench.err
python - $i <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r21/bench{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("value %.3f M" % (d["value"] / 1e6), "ms/step %.4f" % d["ms_per_step"], {k: round(v["ms_total"] / max(v["launches"], 1) * 1e3, 2) for k, v in d["kernel_times_ms"].items()})
PY
done
MTFJSP_LIB=$P/libmtfjsp_grstamp0.so MTFJSP_STAMP_PRINT=1 timeout 300 python bench.py --steps 72 --warmup 36 --min-seconds 0.01 --no-cpu-baseline --no-env-sweep --no-config-legs 2>&1 | grep GR_STAMP | tail -1 > gpurun_out/r21/stamps.txt
cat gpurun_out/r21/stamps.txt | tr '[' '\n' | awk 'NR>1{printf "[%s ", $0}'; echo
MTFJSP_LIB=$P/libmtfjsp_stamp.so MTFJSP_STAMP_PRINT=1 timeout 300 python bench.py --steps 72 --warmup 36 --min-seconds 0.01 --no-cpu-baseline --no-env-sweep --no-config-legs 2>&1 | grep "STAMP" | grep -v "GR_STAMP" | sort | uniq -c | sort -rn | head -12 | cut -c1-330
