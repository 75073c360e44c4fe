#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime sums of the heads / GAT kernels (build variant -DMTFJSP_STAMP; never part of the product build).
    gpurun -- 'python tools/stamp_heads.py'"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mtfjsp_amd  # noqa
from importlib import import_module
b = import_module("e2e-mappo-for-mt-fjsp_amd._build")
lib = b.build_variant("stamp", ["-DMTFJSP_STAMP"] + sys.argv[1:])
env = dict(os.environ, MTFJSP_LIB=lib, MTFJSP_STAMP_PRINT="1")
subprocess.call([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "72", "--warmup", "36", "--min-seconds", "0.01", "--no-cpu-baseline", "--no-env-sweep", "--no-config-legs"], env=env)
