"""Builds libmtfjsp.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libmtfjsp.so")
SOURCES = ["mtfjsp_env.hip", "mtfjsp_encoder.hip", "mtfjsp_gin_res.hip", "mtfjsp_hostgen.cpp"]
# -ffp-contract=off: the scheduling state must follow the reference's binary64 operation order exactly
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall"]
# per translation unit (round 6, A/B on one box with the whole library built either way): hipcc's max-ILP scheduling strategy shortens the step kernels
# (10.3 -> 10.0 us by HIP events at the headline shape, 27.5 -> 26.7 at J10M10) and k_gin_res (108.9 -> 107.6 us), lengthens the streaming product
# kernels (k_gemm_x6: +6 us per launch) and the heads (+0.3 us): the single-launch GIN kernel has a translation unit of its own for that
# (other strategies for k_gin_res: iterative-ilp 110.8, max-memory-clause 108.7, max-ilp with the register-pressure trackers 111.7 us; max-ilp also takes its 12 B
# of scratch per lane to none.)  The rest of the encoder unit keeps the default: without the unclustered high-register-pressure rescheduling stage the three-in-one
# heads launch is 0.6 us shorter but spills 88 instead of 36 B per lane — WRITE_SIZE 3.5 -> 6.9 MB per launch: not taken
SOURCE_FLAGS = {"mtfjsp_env.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"], "mtfjsp_gin_res.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}
if os.environ.get("MTFJSP_NO_SOURCE_FLAGS"):                       # (A/B builds: every translation unit with the common flags)
    SOURCE_FLAGS = {}
if os.environ.get("MTFJSP_SOURCE_FLAGS_JSON"):                     # (A/B builds: {"file": [flags]} replaces the table)
    import json
    SOURCE_FLAGS = json.loads(os.environ["MTFJSP_SOURCE_FLAGS_JSON"])


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps.append(os.path.join(PKG, "..", "include", "mtfjsp.h"))
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile_link(out, extra_flags, verbose=False):
    """one object per source (its own flags, in parallel), then the shared library"""
    import concurrent.futures
    import tempfile
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    with tempfile.TemporaryDirectory(prefix="mtfjsp_build_") as tmp:
        def one(src):
            obj = os.path.join(tmp, src + ".o")
            cmd = [hipcc()] + FLAGS + SOURCE_FLAGS.get(src, []) + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            return obj
        with concurrent.futures.ThreadPoolExecutor(len(srcs)) as ex:
            objs = list(ex.map(one, srcs))
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return out


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    return _compile_link(LIB, [], verbose)


def build_variant(tag, extra_flags, verbose=False):
    """diagnostic builds (never used by the product path): libmtfjsp_<tag>.so, selected with MTFJSP_LIB=<path>"""
    out = os.path.join(PKG, f"libmtfjsp_{tag}.so")
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    # a variant built in the build container ships with the snapshot: rebuilt only when a source or a flag changed (content hash —
    # file times do not survive the copy)
    import hashlib
    hsh = hashlib.sha256(" ".join(FLAGS + list(extra_flags) + [repr(sorted(SOURCE_FLAGS.items()))]).encode())
    for d in sorted(srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))] + [os.path.join(PKG, "..", "include", "mtfjsp.h")]):
        hsh.update(open(d, "rb").read())
    stamp = out + ".srchash"
    if os.path.exists(out) and os.path.exists(stamp) and open(stamp).read() == hsh.hexdigest() and not os.environ.get("MTFJSP_REBUILD_VARIANTS"):
        return out
    _compile_link(out, extra_flags, verbose)
    with open(stamp, "w") as f:
        f.write(hsh.hexdigest())
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
