#!/usr/bin/env python3
"""Diagnostic: phase stamps of the step kernel at a given size (build variant -DMTFJSP_STAMP; never part of the product build).
    gpurun -- 'python tools/stamp_env.py 10x10x2 8192'"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mtfjsp_amd  # noqa
from importlib import import_module
b = import_module("e2e-mappo-for-mt-fjsp_amd._build")
lib = b.build_variant("stamp" + "".join(x.replace("-D", "_") for x in sys.argv[3:]), ["-DMTFJSP_STAMP"] + sys.argv[3:])
env = dict(os.environ, MTFJSP_LIB=lib, MTFJSP_STAMP_PRINT="1")
size, batch = (sys.argv[1:] + ["10x10x2", "8192"])[:2]
subprocess.call([sys.executable, os.path.join(ROOT, "bench.py"), "--size", size, "--batch", batch, "--policy", "random", "--steps", "100", "--warmup", "100",
                 "--min-seconds", "0.01", "--no-cpu-baseline", "--no-env-sweep"], env=env)
