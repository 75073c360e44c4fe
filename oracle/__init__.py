"""CPU oracle for the MT-FJSP hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (e2e-mappo-for-mt-fjsp_amd) never does.
Parity status: pinned against tests/golden/*.npz (captured from the reference).
"""
