#!/bin/bash
# SQ / SQC counters of one kernel of the headline rollout (rocprofv3 --pmc only with --kernel-trace; one counter set per pass)
#   bash tools/pmc_kernel.sh k_headsx "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH"
kern=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out; rm -rf gpurun_out/pmc_k_*
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_k_$i -- python3 bench.py --steps 36 --warmup 36 --min-seconds 0.01 --no-cpu-baseline --no-env-sweep --no-config-legs > /dev/null 2>&1
done
python3 - "$kern" <<'PY'
import csv, glob, collections, sys
kern = sys.argv[1]
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("gpurun_out/pmc_k_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if not name.endswith(kern) and kern not in name: continue
        k = (name[:32], r["Counter_Name"])
        acc[k] += float(r["Counter_Value"]); n[k] += 1
for k in sorted(acc): print(f"{k[0]:34s} {k[1]:28s} {acc[k]/n[k]:16.0f}  (avg per launch over {n[k]})")
PY
