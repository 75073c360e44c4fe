"""No GPU needed: the kernels' gfx950 assembly must not contain the packed-f32 operand swizzle that round 4's microbenchmarks found
unreliable next to matrix instructions (tools/isa_lint.py explains).  Both forms of the heads / GAT kernels are linted:
the textual bodies the product ships and the __forceinline__-function form (-DMTFJSP_BODY_FUNCS=3), whose round-3 schedule contained
six of these instructions per GAT kernel and miscomputed a few node rows per launch."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402


def test_lint_rule_on_samples():
    asm = """
_Z7k_gat3x7GatArgs:
\tv_pk_mul_f32 v[186:187], v[186:187], v[122:123] op_sel:[0,1] op_sel_hi:[1,0]
\tv_pk_mul_f32 v[188:189], v[6:7], v[122:123]
\tv_pk_mul_f32 v[6:7], v[6:7], 0.5 op_sel_hi:[1,0]
\tv_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9] op_sel:[1,0,0]
\tv_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9] op_sel:[0,1,0] op_sel_hi:[1,0,1]
\tv_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel:[1,1] op_sel_hi:[0,0]
\tv_pk_add_u16 v2, v4, v6 op_sel:[0,1]
"""
    bad = isa_lint.lint_asm(asm)
    assert [b[1] for b in bad] == [3, 7] and all(b[0] == "_Z7k_gat3x7GatArgs" for b in bad)


@pytest.mark.parametrize("flags", [(), ("-DMTFJSP_BODY_FUNCS=3",)], ids=["product", "function-form"])
def test_kernels_contain_no_unreliable_packed_f32_swizzle(flags):
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    stream_bad = []
    bad = isa_lint.lint_sources(flags, streaming=stream_bad)
    assert not bad, "\n".join(f"{s}: {k}: {i}" for s, k, _, i in bad)
    # rule 2: the read-once kernels really carry their non-temporal loads (a run-time flag is silently merged into a plain load)
    assert not stream_bad, "\n".join(f"{k}: {w} (found {n})" for k, w, n in stream_bad)


def test_streaming_rule_on_samples():
    asm = """
_Z17k_job_pool_gatherILi1EEvPjiiiiiPKfPKddS2_S2_PKiPfS7_S7_:
\tglobal_load_dwordx4 v[12:15], v[4:5], off
_Z17k_job_pool_gatherILi0EEvPjiiiiiPKfPKddS2_S2_PKiPfS7_S7_:
\tglobal_load_dwordx4 v[12:15], v[4:5], off nt
_Z9k_gemm_x6ILi1EEv8GemmArgs:
\tglobal_load_dwordx4 v[12:15], v[4:5], off nt
"""
    bad = isa_lint.lint_streaming(asm)
    assert sorted(b[0][:28] for b in bad) == ["_Z17k_job_pool_gatherILi0EEv", "_Z17k_job_pool_gatherILi1EEv", "_Z9k_gemm_x6ILi2EE"]
