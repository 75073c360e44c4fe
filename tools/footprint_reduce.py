#!/usr/bin/env python3
"""reduce the per-dispatch traces of tools/footprint_kernel_only.sh (gpurun_out/fp_<tag>_<B>/**/kernel_trace.csv) to
gpurun_out/<tag>_footprint_kernel_only.{json,txt}:  python3 tools/footprint_reduce.py <tag>"""
import csv, collections, glob, json, re, sys
tag = sys.argv[1]
out = {"what": "rocprofv3 --kernel-trace durations (End - Start per dispatch, ns): the step kernel and the SURVEY 8(d) same-footprint copy kernel in one process "
               "(tools/footprint_kernel_only.py); copy: best grid per access width; frac = copy / step",
       "batches": {}}
for B in (4096, 16384, 65536, 262144):
    fs = glob.glob(f"gpurun_out/fp_{tag}_{B}/**/*kernel_trace.csv", recursive=True)
    if not fs:
        continue
    step = collections.defaultdict(list); copy = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        name = r["Kernel_Name"]; dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        gx = int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0); wx = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", "1")) or 1)
        if "k_footprint_copy" in name:
            mm = re.search(r"__vector\((\d)\)", name) or re.search(r"ext_vector_type\((\d)\)", name) or re.search(r"fp_u(\d)", name)
            acc = 4 * int(mm.group(1)) if mm else 4
            copy[(acc, gx // max(wx, 1))].append(dur)
        elif "k_env_grp" in name or "k_env_step" in name or "k_env_reg" in name:
            step[name.split("(")[0].replace("void ", "")].append(dur)
    rec = {}
    for k, v in step.items():
        v = v[len(v) // 3:]                                    # skip the first episode (cold)
        rec["step_kernel"] = k; rec["step_ns_avg"] = sum(v) / len(v); rec["step_launches"] = len(v); rec["step_ns_min"] = min(v)
    for acc in (16, 8):
        best = None
        for (a_, g), v in copy.items():
            if a_ != acc:
                continue
            v = sorted(v)[: max(1, len(v) * 9 // 10)]          # (drop the slowest tenth: the first launches of a group)
            avg = sum(v) / len(v)
            if best is None or avg < best["ns_avg"]:
                best = {"ns_avg": avg, "grid": g, "launches": len(v), "ns_min": v[0]}
        if best:
            rec[f"copy_{acc}B"] = best
    if "step_ns_avg" in rec and "copy_16B" in rec:
        rec["frac_of_same_footprint_copy_kernel_only"] = rec["copy_16B"]["ns_avg"] / rec["step_ns_avg"]
        rec["frac_8B_accesses"] = rec["copy_8B"]["ns_avg"] / rec["step_ns_avg"] if "copy_8B" in rec else None
    out["batches"][str(B)] = rec
json.dump(out, open(f"gpurun_out/{tag}_footprint_kernel_only.json", "w"), indent=1)
with open(f"gpurun_out/{tag}_footprint_kernel_only.txt", "w") as f:
    for B, rec in out["batches"].items():
        line = (f"B={B:>7}  {rec.get('step_kernel','?'):<16} {rec.get('step_ns_avg',0)/1e3:8.2f} us (min {rec.get('step_ns_min',0)/1e3:.2f}, n={rec.get('step_launches',0)})   "
                f"copy16 {rec.get('copy_16B',{}).get('ns_avg',0)/1e3:7.2f} us (grid {rec.get('copy_16B',{}).get('grid')})   copy8 {rec.get('copy_8B',{}).get('ns_avg',0)/1e3:7.2f} us   "
                f"frac {rec.get('frac_of_same_footprint_copy_kernel_only',0):.3f}")
        print(line); f.write(line + "\n")
