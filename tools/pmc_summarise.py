#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) into profiles/<round>_pmc_traffic.json.

    python tools/pmc_summarise.py gpurun_out/r02g_pmc_raw.json profiles/r02_pmc_traffic.json

Input: {kernel name prefix: {FETCH_SIZE_avg, WRITE_SIZE_avg (KiB per launch), launches_*}} as written by the inline
post-processing of the gpurun command.  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE under-reports wide
(16 B/lane) coalesced reads by 2x.
"""
import json
import sys

FAMILY = {
    "void k_gemm_x6<1>": "gin_gemm_bn_relu", "void k_gemm_x6<2>": "gin_gemm_agg", "void k_gemm_x6<3>": "gin0_agg_linear12", "void k_gemm_x6<4>": "gin0_bn_gemm", "k_gin0_moments": "gin0_moments",
    "void k_gemm16p<1>": "gin_gemm_bn_relu", "void k_gemm16p<2>": "gin_gemm_agg", "void k_env_reg<float>": "env_step", "void k_env_grp16<float>": "env_step", "void k_env_grp4<float>": "env_step", "void k_env_grp16x2<float>": "env_step", "void k_env_grp4x2<float>": "env_step", "void k_env_step_grp<float>": "env_step",
    "void k_headsx_gat3x_headsx": "heads_gat3_heads", "k_headsx_gat3x_headsx": "heads_gat3_heads", "k_headsx_values": "heads_values", "k_headsx_gat3x": "heads_gat3", "k_heads": "heads", "k_gat3": "gat3", "void k_gin0<float>": "gin0_agg_linear12", "k_job_pool_gather": "job_pool_gather",
    "void k_mfea1<float>": "mfea1", "k_gin_res": "gin_resident", "void k_env_reset<float>": "env_reset", "k_gae": "gae", "k_snapshot": "snapshot",
}
raw = json.load(open(sys.argv[1]))
out = {"_provenance": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python3 bench.py "
       "--no-cpu-baseline --no-env-sweep --steps 72 --warmup 36 (tools/profile_rollout.sh) ; B=4096 J6M6E2 obs f32; averages per launch. Counters are in KiB. gfx950: "
       "FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced reads (MI355X_MICROARCH.md §HBM) -> "
       "fetch_corrected = 2*FETCH_SIZE; WRITE_SIZE is exact. k_env_reg mixes 4/8/16-B accesses: its read side is "
       "uncalibrated, traffic_bytes uses the 2x (upper) figure and the raw figure is kept beside it.", "kernels": {}}
for name, v in raw.items():
    fam = next((f for k, f in FAMILY.items() if name.startswith(k)), None)
    if fam is None:
        continue
    fr, wr = v["FETCH_SIZE_avg"] * 1024, v["WRITE_SIZE_avg"] * 1024
    out["kernels"][fam] = {"kernel": name.split("(")[0], "fetch_bytes_raw": fr, "fetch_bytes_corrected": 2 * fr, "write_bytes": wr,
                           "traffic_bytes": 2 * fr + wr, "launches": v["launches_fetch"]}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out["kernels"].items():
    print(f"{k:22s} {v['traffic_bytes'] / 1e6:9.2f} MB/launch")
