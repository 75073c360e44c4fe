#!/usr/bin/env python3
"""reduce the per-dispatch traces of tools/footprint_kernel_only.sh (gpurun_out/fp_<tag>_<B>/**/kernel_trace.csv) to
gpurun_out/<tag>_footprint_kernel_only.{json,txt}:  python3 tools/footprint_reduce.py <tag> [dir-prefix]

tools/footprint_kernel_only.py runs episode 0 as warm-up, odd episodes in the rollout's launch pattern (action kernel, step kernel,
back to back) and even episodes >= 2 with every step launch between two HIP event records, which is how every timed copy launch is
made.  Like is compared with like:
  frac_of_same_footprint_copy_kernel_only = copy (between events) / step (between events)       <- the figure to quote
  frac_round5_method                      = copy (between events) / step (rollout pattern)      <- round 5's quotient, kept for the history
  frac_b2b                                = copy (warm-up launches, queued back to back) / step (rollout pattern)
(episodes are told apart by the k_env_reset dispatches in the trace)"""
import csv, collections, glob, json, re, sys
tag = sys.argv[1]
prefix = sys.argv[2] if len(sys.argv) > 2 else "fp"
size = sys.argv[3] if len(sys.argv) > 3 else "6x6x2"                # what tools/footprint_kernel_only.py ran (--size)
out = {"size": size, "what": "rocprofv3 --kernel-trace durations (End - Start per dispatch, ns) of the step kernel and of the SURVEY 8(d) same-footprint copy kernel in one "
               "process (tools/footprint_kernel_only.py), by the gap in front of the dispatch: idle (>= 1 us) | b2b (< 0.2 us); copy: best grid per class",
       "batches": {}}


def stats(v):
    v = sorted(v)
    return {"ns_avg": sum(v) / len(v), "ns_median": v[len(v) // 2], "ns_min": v[0], "launches": len(v)} if v else None


for B in (4096, 16384, 65536, 262144):
    fs = glob.glob(f"gpurun_out/{prefix}_{tag}_{B}/**/*kernel_trace.csv", recursive=True)
    if not fs:
        continue
    rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
    step = {"events": [], "stream": []}
    copy = collections.defaultdict(lambda: {"idle": [], "b2b": []})
    prev_end, ep, kname = None, -1, "?"
    for r in rows:
        name = r["Kernel_Name"]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) if prev_end is not None else 10 ** 9
        prev_end = max(e, prev_end or 0)
        gx = int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0); wx = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", "1")) or 1)
        if "k_env_reset" in name:
            ep += 1
        elif "k_footprint_copy" in name:
            mm = re.search(r"__vector\((\d)\)", name) or re.search(r"ext_vector_type\((\d)\)", name) or re.search(r"fp_u(\d)", name)
            acc = 4 * int(mm.group(1)) if mm else 4
            cls = "idle" if gap >= 1000 else "b2b" if gap < 200 else None
            if cls:
                copy[(acc, gx // max(wx, 1))][cls].append(e - s)
        elif "k_env_grp" in name or "k_env_step" in name or "k_env_reg" in name:
            kname = name.split("(")[0].replace("void ", "")
            if ep >= 1:
                step["events" if (ep >= 2 and ep % 2 == 0) else "stream"].append(e - s)
    rec = {"step_kernel": kname, "step_idle": stats(step["events"]), "step_b2b": stats(step["stream"])}
    for acc in (16, 8):
        for cls in ("idle", "b2b"):
            best = None
            for (a_, g), v in copy.items():
                if a_ != acc or len(v[cls]) < 5:
                    continue
                vv = sorted(v[cls])[: max(1, len(v[cls]) * 9 // 10)]          # (drop the slowest tenth)
                avg = sum(vv) / len(vv)
                if best is None or avg < best["ns_avg"]:
                    best = {"ns_avg": avg, "grid": g, "launches": len(vv), "ns_min": vv[0]}
            rec[f"copy_{acc}B_{cls}"] = best
    ci, cb = rec.get("copy_16B_idle"), rec.get("copy_16B_b2b")
    if rec["step_idle"] and ci:
        rec["frac_of_same_footprint_copy_kernel_only"] = ci["ns_avg"] / rec["step_idle"]["ns_avg"]
    if rec["step_b2b"] and cb:
        rec["frac_b2b"] = cb["ns_avg"] / rec["step_b2b"]["ns_avg"]
    if rec["step_b2b"] and ci:
        rec["frac_round5_method"] = ci["ns_avg"] / rec["step_b2b"]["ns_avg"]
    out["batches"][str(B)] = rec
json.dump(out, open(f"gpurun_out/{tag}_footprint_kernel_only.json", "w"), indent=1)
with open(f"gpurun_out/{tag}_footprint_kernel_only.txt", "w") as f:
    f.write("# step kernel | same-footprint copy, rocprofv3 --kernel-trace durations by what was in front of the dispatch (tools/footprint_reduce.py)\n")
    for B, rec in out["batches"].items():
        def g(k, kk="ns_avg"):
            return (rec.get(k) or {}).get(kk, 0) / 1e3
        line = (f"B={B:>7}  {rec.get('step_kernel','?'):<18} between event records: step {g('step_idle'):7.2f} us (min {g('step_idle','ns_min'):.2f}, n={(rec.get('step_idle') or {}).get('launches',0)})"
                f"  copy16 {g('copy_16B_idle'):7.2f} us  copy8 {g('copy_8B_idle'):7.2f} us  frac {rec.get('frac_of_same_footprint_copy_kernel_only',0):.3f}"
                f"   |  rollout pattern: step {g('step_b2b'):7.2f} us (n={(rec.get('step_b2b') or {}).get('launches',0)})  copy16 {g('copy_16B_b2b'):7.2f} us  frac {rec.get('frac_b2b',0):.3f}"
                f"   |  round-5 method (copy between events / step in the rollout pattern) {rec.get('frac_round5_method',0):.3f}")
        print(line); f.write(line + "\n")
