#!/bin/bash
# A/B of step-kernel builds on ONE box: rocprofv3 --kernel-trace durations of the step kernel (and of the same-footprint copy) for each
# library in turn, repeated (devices of the pool differ by 5-10 %: only same-box numbers compare).
#   gpurun -- 'bash tools/ab_env_kernel.sh r06 3 4096 libmtfjsp_ab_r05.so libmtfjsp.so'
set -u
tag=$1; reps=$2; B=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
for r in $(seq 1 $reps); do
  for lib in "$@"; do
    name=$(basename $lib .so)
    MTFJSP_LIB=$PWD/e2e-mappo-for-mt-fjsp_amd/$lib timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ab_${tag}_${name}_$r -- python3 tools/footprint_kernel_only.py --batch $B > gpurun_out/ab_${tag}_${name}_$r.log 2>&1
  done
done
python3 - "$tag" "$reps" "$@" <<'PY' | tee gpurun_out/ab_${1}_summary.txt
import csv, glob, sys, collections
tag, reps, libs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
for lib in libs:
    name = lib[:-3]
    for r in range(1, reps + 1):
        fs = glob.glob(f"gpurun_out/ab_{tag}_{name}_{r}/**/*kernel_trace.csv", recursive=True)
        if not fs:
            print(name, r, "no trace"); continue
        step, copy = [], collections.defaultdict(list)
        for row in csv.DictReader(open(fs[0])):
            dur = float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); k = row["Kernel_Name"]
            if "k_footprint_copy" in k: copy[(k, row.get("Grid_Size_X", row.get("Grid_Size")))].append(dur)
            elif "k_env_grp" in k or "k_env_step" in k or "k_env_reg" in k: step.append(dur)
        step = step[len(step) // 3:]
        best = min(sum(sorted(v)[:max(1, len(v) * 9 // 10)]) / max(1, len(v) * 9 // 10) for v in copy.values()) if copy else float("nan")
        s = sorted(step)
        print(f"{name:28s} rep {r}: step avg {sum(step)/len(step)/1e3:6.2f} us  median {s[len(s)//2]/1e3:6.2f}  min {s[0]/1e3:6.2f}  (n={len(step)})   copy {best/1e3:5.2f} us   frac {best/(sum(step)/len(step)):.3f}")
PY
