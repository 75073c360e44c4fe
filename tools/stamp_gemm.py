#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime sums of the streaming GIN launches at a given size (build variant -DMTFJSP_STAMP).
    gpurun -- 'python tools/stamp_gemm.py 10x10x2 8192'"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mtfjsp_amd  # noqa
from importlib import import_module
b = import_module("e2e-mappo-for-mt-fjsp_amd._build")
lib = b.build_variant("stamp" + "".join(x.replace("-D", "_") for x in sys.argv[3:]), ["-DMTFJSP_STAMP"] + sys.argv[3:])
env = dict(os.environ, MTFJSP_LIB=lib, MTFJSP_STAMP_PRINT="1")
size, batch = (sys.argv[1:] + ["10x10x2", "8192"])[:2]
subprocess.call([sys.executable, os.path.join(ROOT, "bench.py"), "--size", size, "--batch", batch, "--steps", "20", "--warmup", "5", "--min-warmup-seconds", "0",
                 "--min-seconds", "0.01", "--no-cpu-baseline", "--no-env-sweep"], env=env)
