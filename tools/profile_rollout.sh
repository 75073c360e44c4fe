#!/bin/bash
# The three profiling passes behind profiles/<tag>_*: bench JSON, rocprofv3 kernel-trace stats of the same command, and the
# HBM traffic counters (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, never combined with other trace domains).
#   gpurun -- 'bash tools/profile_rollout.sh r01n'   ->  gpurun_out/<tag>_bench.json, <tag>_kernel_stats.csv, <tag>_pmc_raw.json
set -u
tag=${1:-prof}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python3 bench.py --no-cpu-baseline --no-env-sweep --no-config-legs > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_${tag} -- python3 bench.py --no-cpu-baseline --no-env-sweep --no-config-legs --steps 72 --warmup 36 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_${tag} -- python3 bench.py --no-cpu-baseline --no-env-sweep --no-config-legs --steps 72 --warmup 36 > /dev/null 2>&1
python3 - "$tag" <<'PY'
import csv, collections, glob, json, shutil, sys
tag = sys.argv[1]
out = {}
for name in ("fetch", "write"):
    fs = glob.glob(f"gpurun_out/pmc_{name}_{tag}/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:60]
        acc[k] += float(r["Counter_Value"]); n[k] += 1
    for k in acc:
        if k.startswith("void k_") or k.startswith("k_"):
            out.setdefault(k, {})[name.upper() + "_SIZE_avg"] = acc[k] / n[k]; out[k]["launches_" + name] = n[k]
json.dump(out, open(f"gpurun_out/{tag}_pmc_raw.json", "w"), indent=1)
ks = glob.glob(f"gpurun_out/prof_{tag}/**/*kernel_stats.csv", recursive=True)
if ks:
    shutil.copy(ks[0], f"gpurun_out/{tag}_kernel_stats.csv")
    for line in open(ks[0]).read().splitlines()[:12]:
        print(line[:110])
PY
