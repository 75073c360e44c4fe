#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_env_hip_golden.py tests/test_env_edge_cases_gpu.py tests/test_fused_env_step_gpu.py tests/test_parallel_env_dropin_gpu.py -q -m gpu -x 2>&1 | tail -5
python3 bench.py --no-cpu-baseline > gpurun_out/envb.json 2> gpurun_out/envb.err; tail -3 gpurun_out/envb.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/envb.json").read().strip().splitlines()[-1])
print("value %.3f M" % (d["value"] / 1e6), "ms/step %.4f" % d["ms_per_step"], {k: round(v["ms_total"] / max(v["launches"], 1) * 1e3, 2) for k, v in d["kernel_times_ms"].items()})
for s in d.get("roofline_env_step_batch_sweep", []):
    print(s["instances"], s["kernel"], "%.2f us" % s["avg_launch_us"], "frac %.3f / %.3f" % (s["frac_of_same_footprint_copy"], s["frac_of_same_footprint_copy_8B_accesses"]))
print({k: d[k] for k in d if k.startswith("roofline_heads") or k.startswith("trajectory")})
print(d.get("roofline_env_step"))
PY
python - <<'PY'
import json
d = json.loads(open("gpurun_out/envb.json").read().strip().splitlines()[-1])
for k, v in d.get("configs", {}).items():
    print(k, "%.3f M" % (v.get("value", 0) / 1e6), v.get("kernel_times_us_per_launch", {}).get("env_step"))
PY
