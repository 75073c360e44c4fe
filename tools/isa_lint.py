#!/usr/bin/env python3
"""ISA lint of the HIP sources (no GPU needed): compiles every kernel source to gfx950 assembly and refuses instruction forms that
this round's microbenchmarks showed to be unreliable on MI355X.

Rule 1 — packed-f32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) whose LOW result reads the HIGH half of source 1
while reading the low half of source 0 (`op_sel:[0,1...]`): with a second wave on the SIMD executing matrix instructions, lanes
48..63 of about 0.3 % of the executions are wrong (tools/ubench/valu_after_mfma.hip; profiles/r04_ubench_valu_after_mfma.txt).
hipcc's SLP vectoriser produces the form when it pairs scalar products whose second operands sit in a register pair in the opposite
order.  Every other op_sel combination measured (source 0 swapped, both swapped, low-twice, source-0 / source-2 high-twice) is clean.

Rule 2 — the kernels that read a matrix ONCE must do it with non-temporal loads (`global_load_dwordx4 ... nt`): a 419 MB matrix
just written by the previous launch reads at 3.6 TB/s with plain loads and at 5.0-5.2 TB/s with non-temporal ones
(tools/ubench/read_after_write.hip).  A run-time `nt ? __builtin_nontemporal_load(p) : *p` is merged by the optimiser into ONE
plain load — k_job_pool_gather shipped like that for two rounds — so the expectation is checked on the ISA: STREAMING lists the
kernels (mangled-name prefixes) and whether their ISA must / must not contain such loads.

    python tools/isa_lint.py [extra hipcc flags ...]      exit code 1 and a listing when a rule fires
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "e2e-mappo-for-mt-fjsp_amd", "csrc")
SOURCES = ["mtfjsp_env.hip", "mtfjsp_encoder.hip", "mtfjsp_gin_res.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-S", "--cuda-device-only", "-w"]
PK_F32 = re.compile(r"^\s*(v_pk_(?:mul|add|fma)_f32)\b(.*)$")
OP_SEL = re.compile(r"op_sel:\[([01]),([01])")


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def lint_asm(text):
    """-> list of (kernel, line number, instruction) violating rule 1"""
    bad, kernel = [], None
    for n, line in enumerate(text.splitlines(), 1):
        m = re.match(r"^(_Z\w+|k_\w+):", line)
        if m:
            kernel = m.group(1)
            continue
        m = PK_F32.match(line)
        if not m:
            continue
        o = OP_SEL.search(m.group(2))
        if o and o.group(1) == "0" and o.group(2) == "1":
            bad.append((kernel, n, line.strip()))
    return bad


NT_LOAD = re.compile(r"^\s*global_load_dwordx4\b.*\bnt\b")
# (mangled-name prefix, must contain non-temporal 16-byte loads?)
STREAMING = [("_Z17k_job_pool_gatherILi1EE", True), ("_Z17k_job_pool_gatherILi0EE", False),
             ("_Z9k_gemm_x6ILi1EE", True), ("_Z9k_gemm_x6ILi2EE", True)]


def nt_load_counts(text):
    """-> {kernel: number of non-temporal 16-byte global loads in its ISA}"""
    counts, kernel = {}, None
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+|k_\w+):", line)
        if m:
            kernel = m.group(1)
            counts.setdefault(kernel, 0)
        elif kernel and NT_LOAD.match(line):
            counts[kernel] += 1
    return counts


def lint_streaming(text):
    """-> list of (kernel prefix, expectation, count) for the STREAMING kernels whose ISA disagrees with the expectation"""
    counts = nt_load_counts(text)
    bad = []
    for prefix, want in STREAMING:
        ks = [k for k in counts if k.startswith(prefix)]
        for k in ks:
            if (counts[k] > 0) != want:
                bad.append((k, "non-temporal loads expected" if want else "no non-temporal loads expected", counts[k]))
        if not ks:
            bad.append((prefix, "kernel not found in the assembly", 0))
    return bad


def compile_asm(extra_flags=()):
    """-> [(source file, gfx950 assembly text)]"""
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for src in SOURCES:
            asm = os.path.join(tmp, src + ".s")
            subprocess.check_call([hipcc()] + FLAGS + list(extra_flags) + [os.path.join(CSRC, src), "-o", asm], stderr=subprocess.DEVNULL)
            out.append((src, open(asm).read()))
    return out


def lint_sources(extra_flags=(), streaming=None):
    """rule 1 over every source; with `streaming` (a list) rule 2's findings for mtfjsp_encoder.hip are appended to it"""
    out = []
    for src, text in compile_asm(extra_flags):
        out += [(src,) + b for b in lint_asm(text)]
        if streaming is not None and src == "mtfjsp_encoder.hip":
            streaming += lint_streaming(text)
    return out


if __name__ == "__main__":
    stream_bad = []
    bad = lint_sources(sys.argv[1:], streaming=stream_bad)
    for src, kernel, n, ins in bad:
        print(f"{src}: {kernel}: line {n}: {ins}")
    for kernel, what, n in stream_bad:
        print(f"mtfjsp_encoder.hip: {kernel}: {what} (found {n})")
    print(f"isa_lint: {len(bad)} unreliable packed-f32 operand swizzle(s), {len(stream_bad)} streaming-load mismatch(es)" + (" with " + " ".join(sys.argv[1:]) if sys.argv[1:] else ""))
    sys.exit(1 if bad or stream_bad else 0)
