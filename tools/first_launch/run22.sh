#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r22
timeout 1500 python -m pytest tests/test_encoder_hip.py tests/test_full_size_gpu.py tests/test_encoder_sizes_gpu.py tests/test_first_launch_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python bench.py --no-env-sweep --no-cpu-baseline > gpurun_out/r22/bench.json 2> gpurun_out/r22/bench.err
MTFJSP_POOL_OLD=1 timeout 600 python bench.py --no-env-sweep --no-cpu-baseline > gpurun_out/r22/bench_poolold.json 2>> gpurun_out/r22/bench.err
python - <<'PY'
import json
for f in ("bench", "bench_poolold"):
    d = json.loads(open(f"gpurun_out/r22/{f}.json").read().strip().splitlines()[-1])
    print(f, "value %.3f M" % (d["value"] / 1e6), "ms/step %.4f" % d["ms_per_step"], {k: round(v["ms_total"] / max(v["launches"], 1) * 1e3, 2) for k, v in d["kernel_times_ms"].items()})
    for k, v in d.get("configs", {}).items():
        print("   ", k, "%.3f M" % (v.get("value", 0) / 1e6), {a: round(b, 1) for a, b in v.get("kernel_times_us_per_launch", {}).items()})
PY
