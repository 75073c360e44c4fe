#!/usr/bin/env python3
"""Diagnostic: phase stamps of the resident GIN kernel (build variant -DGR_STAMP; never part of the product build).
    gpurun -- 'python tools/stamp_gin_resident.py'"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mtfjsp_amd  # noqa
from importlib import import_module
b = import_module("e2e-mappo-for-mt-fjsp_amd._build")
for abl in [int(x) for x in (sys.argv[1:] or ["0"])]:
    extra = os.environ.get("GR_EXTRA_FLAGS", "").split()           # e.g. GR_EXTRA_FLAGS="-DGR_FLATBAR"
    lib = b.build_variant(f"grstamp{abl}" + "".join(x.replace("-D", "_") for x in extra), ["-DGR_STAMP", f"-DGR_ABL={abl}"] + extra)
    env = dict(os.environ, MTFJSP_LIB=lib, MTFJSP_STAMP_PRINT="1")
    print("GR_ABL", abl, flush=True)
    subprocess.call([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "72", "--warmup", "36", "--min-seconds", "0.01", "--no-cpu-baseline", "--no-env-sweep"], env=env)
