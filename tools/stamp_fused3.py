#!/usr/bin/env python3
"""Diagnostic: s_memrealtime stamps of the three-in-one heads launch (build variant -DMTFJSP_STAMP3; never part of the product build).
    gpurun -- 'python tools/stamp_fused3.py'"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mtfjsp_amd  # noqa
from importlib import import_module
b = import_module("e2e-mappo-for-mt-fjsp_amd._build")
lib = b.build_variant("stamp3", ["-DMTFJSP_STAMP3"] + sys.argv[1:])
env = dict(os.environ, MTFJSP_LIB=lib, MTFJSP_STAMP_PRINT="1")
subprocess.call([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "72", "--warmup", "36", "--min-seconds", "0.01", "--no-cpu-baseline", "--no-env-sweep", "--no-config-legs"], env=env)
