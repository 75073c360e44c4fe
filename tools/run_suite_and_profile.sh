#!/bin/bash
# full GPU suite, then the three profiling passes of the headline bench (tools/profile_rollout.sh <tag>)
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r04a}
rm -f gpurun_out/encoder_errors_observed.json
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -12
bash tools/profile_rollout.sh $tag 2>&1 | tail -14
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
d = json.loads(open(f"gpurun_out/{tag}_bench.json").read().strip().splitlines()[-1])
print("value %.3f M" % (d["value"] / 1e6), "ms/step %.4f" % d["ms_per_step"], {k: round(v["ms_total"] / max(v["launches"], 1) * 1e3, 2) for k, v in d["kernel_times_ms"].items()})
for s in d.get("roofline_env_step_batch_sweep", []):
    print(s["instances"], s["kernel"], "%.2f us" % s["avg_launch_us"], "copy 16B %.2f us (grid %d) 8B %.2f us -> frac %.3f / %.3f" % (
        s["same_footprint_copy"]["access_16B"]["avg_launch_us"], s["same_footprint_copy"]["access_16B"]["grid"], s["same_footprint_copy"]["access_8B"]["avg_launch_us"],
        s["frac_of_same_footprint_copy"], s["frac_of_same_footprint_copy_8B_accesses"]))
for k, v in d.get("configs", {}).items():
    print(k, "%.3f M" % (v.get("value", 0) / 1e6))
PY
