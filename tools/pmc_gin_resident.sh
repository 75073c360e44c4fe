#!/bin/bash
# SQ counters of the resident GIN kernel (rocprofv3 --pmc only with --kernel-trace; no other trace domains)
#   bash tools/pmc_gin_resident.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES ..." ["second set" ...]
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_gr_$i -- python3 bench.py --steps 36 --warmup 36 --min-seconds 0.01 --no-cpu-baseline --no-env-sweep --no-config-legs > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("gpurun_out/pmc_gr_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_gin_res" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc): print(f"{k:28s} {acc[k]/n[k]:16.0f}  (avg per launch over {n[k]})")
PY
