#!/bin/bash
# Memory-path counters of the streaming launches at one size (rocprofv3 --pmc only with --kernel-trace; one counter set per pass).
#   gpurun -- 'bash tools/pmc_streaming_kernels.sh r06 10x10x2 8192'  ->  gpurun_out/<tag>_pmc_mempath_<size>.txt
tag=${1:-pmc}; size=${2:-10x10x2}; batch=${3:-8192}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
i=0
# (a GRBM_GUI_ACTIVE / TA_* pass hangs rocprofv3 on this pool: left out)
for set in "TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_WRREQ_STALL_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  t0=$(date +%s)
  timeout 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcm_${tag}_$i -- python3 bench.py --size $size --batch $batch --steps 8 --warmup 4 --min-seconds 0.01 --min-warmup-seconds 0 --no-cpu-baseline --no-env-sweep > /dev/null 2>&1
  echo "pass $i ($set): rc=$? $(( $(date +%s) - t0 )) s" >> gpurun_out/${tag}_pmc_mempath_passes.log
done
python3 - "$tag" "$size" "$batch" <<'PY'
import csv, glob, collections, sys
tag, size, batch = sys.argv[1:4]
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(f"gpurun_out/pmcm_{tag}_*/**/*counter_collection.csv", recursive=True):
    seen = collections.Counter()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if "k_gemm_x6" not in name and "moments" not in name and "pool_gather" not in name: continue
        k = (name[:28], r["Counter_Name"])
        acc[k] += float(r["Counter_Value"]); seen[(k, r["Dispatch_Id"])] += 1
    for (k, d) in seen: n[k] += 1
with open(f"gpurun_out/{tag}_pmc_mempath_{size}_B{batch}.txt", "w") as o:
    for k in sorted(acc):
        line = f"{k[0]:30s} {k[1]:40s} {acc[k]/max(n[k],1):18.0f}  (avg per launch over {n[k]})"
        print(line); o.write(line + "\n")
PY
rm -rf gpurun_out/pmcm_${tag}_*
