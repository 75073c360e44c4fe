#!/bin/bash
# HBM traffic counters of the streaming-GIN shapes (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, only with --kernel-trace).
#   gpurun -- 'bash tools/pmc_sizes.sh r03d 20x20x4 2048'  ->  gpurun_out/<tag>_pmc_<size>.txt
tag=${1:-pmc}; size=${2:-20x20x4}; batch=${3:-2048}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmcs_${tag}_$c -- python3 bench.py --size $size --batch $batch --steps 30 --warmup 10 --min-seconds 0.01 --min-warmup-seconds 0 --no-cpu-baseline --no-env-sweep > /dev/null 2>&1
done
python3 - "$tag" "$size" "$batch" <<'PY'
import csv, glob, collections, sys
tag, size, batch = sys.argv[1:4]
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(f"gpurun_out/pmcs_{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])
        acc[k] += float(r["Counter_Value"]); n[k] += 1
with open(f"gpurun_out/{tag}_pmc_{size}_B{batch}.txt", "w") as o:
    for k in sorted(acc):
        if n[k] < 20: continue
        line = f"{k[0]:42s} {k[1]:12s} {acc[k]/n[k]/1024:10.1f} MiB per launch (KiB counter avg over {n[k]}; FETCH_SIZE x2 for 16-B/lane reads on gfx950)"
        print(line); o.write(line + "\n")
PY
