#!/usr/bin/env python3
"""One `configs` leg of bench.py on its own (J10M10E2 x 8192, J20M20E4 x 2048 by default): value, ms per step, HIP-event kernel times.
    gpurun -- 'MTFJSP_FUSE_PAIR=0 python tools/bench_config_leg.py; python tools/bench_config_leg.py'"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

bench.torch = torch
from importlib import import_module  # noqa: E402

ro_mod = import_module("e2e-mappo-for-mt-fjsp_amd.rollout")
sizes = [tuple(int(x) for x in s.split("x")) for s in sys.argv[1:]] or [(10, 10, 2, 8192), (20, 20, 4, 2048)]
for (J, M, E, B) in sizes:
    r = bench.config_leg(ro_mod, J, M, E, B, 0)
    print(json.dumps({"size": [J, M, E, B], "fuse_pair": os.environ.get("MTFJSP_FUSE_PAIR", "default"), "M_env_steps_per_s": round(r["value"] / 1e6, 3),
                      "ms_per_step": round(r["ms_per_step"], 4), "kernel_us": {k: round(v, 1) for k, v in r["kernel_times_us_per_launch"].items()}}), flush=True)
