"""Builds libmtfjsp.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libmtfjsp.so")
SOURCES = ["mtfjsp_env.hip", "mtfjsp_encoder.hip", "mtfjsp_hostgen.cpp"]
# -ffp-contract=off: the scheduling state must follow the reference's binary64 operation order exactly
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-Wall"]


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


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [hipcc()] + FLAGS + srcs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_variant(tag, extra_flags, verbose=False):
    """diagnostic builds (never used by the product path): libmtfjsp_<tag>.so, selected with MTFJSP_LIB=<path>"""
    out = os.path.join(PKG, f"libmtfjsp_{tag}.so")
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    # a variant built in the build container ships with the snapshot: rebuilt only when a source or a flag changed (content hash —
    # file times do not survive the copy)
    import hashlib
    hsh = hashlib.sha256(" ".join(FLAGS + list(extra_flags)).encode())
    for d in sorted(srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))] + [os.path.join(PKG, "..", "include", "mtfjsp.h")]):
        hsh.update(open(d, "rb").read())
    stamp = out + ".srchash"
    if os.path.exists(out) and os.path.exists(stamp) and open(stamp).read() == hsh.hexdigest() and not os.environ.get("MTFJSP_REBUILD_VARIANTS"):
        return out
    cmd = [hipcc()] + FLAGS + list(extra_flags) + srcs + ["-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(hsh.hexdigest())
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
