#!/usr/bin/env python3
"""Diagnostic: thread-scaling curve of the CPU oracle port on this host (no GPU, no torch).  python tools/cpu_scaling_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mtfjsp_amd  # noqa
from importlib import import_module
inst = import_module("e2e-mappo-for-mt-fjsp_amd.instances")
from oracle.env_oracle import OracleBatch, max_threads
print("loadavg", open("/proc/loadavg").read().strip(), "omp", {k: os.environ.get(k) for k in ("OMP_PLACES", "OMP_PROC_BIND", "OMP_NUM_THREADS")}, "max_threads", max_threads())
B = 4096
t, p, tt, edge = inst.generate_instances(B, 6, 6, 2, seed=0)
w3 = np.full((B, 3), 1 / 3)
o = OracleBatch(t, p, tt, edge)
for th in [int(x) for x in (sys.argv[1:] or ["1", "8", "32", "64", "128", "256"])]:
    for ep in (20, 200):
        n, wall, steps = o.bench_blocks(ep, th, w3)
        print(f"threads {th:4d} episodes {ep:4d}: {n / wall / 1e6:8.2f} M env-steps/s  ({n / wall / 1e6 / th:.3f} per thread)", flush=True)
