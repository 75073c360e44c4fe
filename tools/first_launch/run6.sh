#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6
P=e2e-mappo-for-mt-fjsp_amd
timeout 300 tools/ubench/mfma_raw > gpurun_out/r6/mfma_raw.txt 2>&1
cat gpurun_out/r6/mfma_raw.txt
timeout 1500 python -m pytest tests/test_encoder_hip.py tests/test_full_size_gpu.py -x -q -m gpu -k "encoder or resident or actor or activation" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_rollout_handoff.py -x -q -m gpu -k "timeout or drops" 2>&1 | tail -3
timeout 300 python bench.py --no-config-legs --no-env-sweep --no-cpu-baseline > gpurun_out/r6/bench.json 2> gpurun_out/r6/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r6/bench.json").read().strip().splitlines()[-1])
    print("value %.3f M" % (d["value"] / 1e6), "ms/step %.4f" % d["ms_per_step"], {k: round(v["ms_total"] / max(v["launches"], 1) * 1e3, 2) for k, v in d["kernel_times_ms"].items()})
except Exception as ex:
    print("bench failed", ex)
PY
MTFJSP_LIB=$P/libmtfjsp_grstamp0.so MTFJSP_STAMP_PRINT=1 timeout 300 python bench.py --steps 72 --warmup 36 --min-seconds 0.01 --no-cpu-baseline --no-env-sweep --no-config-legs 2>&1 | grep GR_STAMP | tail -2 > gpurun_out/r6/stamps.txt
cut -c1-700 gpurun_out/r6/stamps.txt
