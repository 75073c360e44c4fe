#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r8
timeout 600 tools/ubench/mfma_raw > gpurun_out/r8/mfma_raw.txt 2>&1
grep "v_pk_mul" gpurun_out/r8/mfma_raw.txt | cut -c1-190
timeout 600 python bench.py > gpurun_out/r8/bench.json 2> gpurun_out/r8/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r8/bench.json").read().strip().splitlines()[-1])
print("value %.3f M" % (d["value"] / 1e6), "ms/step %.4f" % d["ms_per_step"], {k: round(v["ms_total"] / max(v["launches"], 1) * 1e3, 2) for k, v in d["kernel_times_ms"].items()})
for s in d.get("roofline_env_step_batch_sweep", []):
    print(s["instances"], s["kernel"], "%.2f us" % s["avg_launch_us"], "frac copy(6.29) %.3f" % s["frac_of_measured_copy_bw"], "same-footprint copy 16B %.2f us (grid %d) 8B %.2f us -> frac %.3f / %.3f" % (
        s["same_footprint_copy"]["access_16B"]["avg_launch_us"], s["same_footprint_copy"]["access_16B"]["grid"], s["same_footprint_copy"]["access_8B"]["avg_launch_us"],
        s["frac_of_same_footprint_copy"], s["frac_of_same_footprint_copy_8B_accesses"]))
for k, v in d.get("configs", {}).items():
    print(k, "%.3f M" % (v.get("value", 0) / 1e6), v.get("kernel_times_us_per_launch"))
print("cpu", d["cpu_baseline"]["value"] / 1e6, d["cpu_baseline"]["cores"])
PY
timeout 900 python tools/stamp_heads.py 2>&1 | grep STAMP | tail -6 | cut -c1-400
