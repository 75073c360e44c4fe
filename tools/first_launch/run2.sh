#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2
P=e2e-mappo-for-mt-fjsp_amd
tools/ubench/split_check > gpurun_out/r2/split_check.txt 2>&1
timeout 600 python tools/first_launch/warm.py $P/libmtfjsp.so 4 fused > gpurun_out/r2/warm_textual.txt 2>&1
for v in fgat fheads funcs; do
  timeout 600 python tools/first_launch/warm.py $P/libmtfjsp_$v.so 4 fused > gpurun_out/r2/warm_${v}_fused.txt 2>&1
  timeout 600 python tools/first_launch/warm.py $P/libmtfjsp_$v.so 4 unfused > gpurun_out/r2/warm_${v}_unfused.txt 2>&1
done
tail -n 6 gpurun_out/r2/*.txt
timeout 1500 python -m pytest tests/test_encoder_hip.py tests/test_full_size_gpu.py tests/test_first_launch_gpu.py -x -q -m gpu 2>&1 | tail -8
timeout 300 python bench.py --no-config-legs --no-env-sweep --no-cpu-baseline > gpurun_out/r2/bench_poll0.json 2> gpurun_out/r2/bench_poll0.err
MTFJSP_GIN_RES_POLL=1 timeout 300 python bench.py --no-config-legs --no-env-sweep --no-cpu-baseline > gpurun_out/r2/bench_poll1.json 2> gpurun_out/r2/bench_poll1.err
python - <<'PY'
import json
for f in ("poll0", "poll1"):
    try:
        d = json.loads(open(f"gpurun_out/r2/bench_{f}.json").read().strip().splitlines()[-1])
        print(f, "value %.3f M" % (d["value"] / 1e6), "ms/step %.4f" % d["ms_per_step"], {k: round(v["ms_total"] / max(v["launches"], 1) * 1e3, 2) for k, v in d["kernel_times_ms"].items()})
    except Exception as ex:
        print(f, "failed", ex)
PY
timeout 900 python -m pytest tests/test_rollout_handoff.py -x -q -m gpu 2>&1 | tail -5
