"""mtfjsp-hip: MI355X-native batched MT-FJSP disjunctive-graph environment + MAPPO rollout.

Host side of the C ABI declared in include/mtfjsp.h.  The compute path is
libmtfjsp.so (hand-written HIP for gfx950); there is NO CPU fallback — importing
`capi` raises if the library has not been built (`python __graft_entry__.py` or
`python e2e-mappo-for-mt-fjsp_amd/_build.py`).

The directory name contains '-' and is therefore imported through importlib:
    import importlib; pkg = importlib.import_module("e2e-mappo-for-mt-fjsp_amd")
or via the root-level alias module `mtfjsp_amd`.
"""
__all__ = ["capi", "batch_env", "parallel_env", "instances", "encoder", "rollout", "dist"]
