#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 2400 python -m pytest tests/test_trajectory_buffer.py tests/test_rollout_handoff.py -q -m gpu -x 2>&1 | tail -8
python3 bench.py --no-cpu-baseline --no-env-sweep > gpurun_out/trajb.json 2> gpurun_out/trajb.err; tail -3 gpurun_out/trajb.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/trajb.json").read().strip().splitlines()[-1])
print("value %.3f M" % (d["value"] / 1e6), "ms/step %.4f" % d["ms_per_step"])
print(d.get("trajectory_full"))
PY
