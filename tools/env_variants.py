#!/usr/bin/env python3
"""Diagnostic: the step kernel alone at several batch sizes for each kernel that can serve J6M6 (MTFJSP_ENV_KERNEL override).
    gpurun -- 'python tools/env_variants.py'"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = """
import sys; sys.path.insert(0, %r)
import torch, bench
out = [bench.env_kernel_large_batch(6, 6, 2, 0, B=B, episodes=1 if B >= 65536 else 2) for B in (4096, 16384, 65536, 262144)]
print("ENVSWEEP", " | ".join("%%d: %%.1f us %%.3f" %% (x["instances"], x["avg_launch_us"], x["frac_of_measured_copy_bw"]) for x in out))
""" % ROOT
for k in (sys.argv[1:] or ["grp16", "grp4", "reg1", "lds"]):
    print("KERNEL", k, flush=True)
    subprocess.call([sys.executable, "-c", code], env=dict(os.environ, MTFJSP_ENV_KERNEL=k))
