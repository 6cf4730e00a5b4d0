"""Builds libnsdg.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU; the built library is git-ignored but travels to the GPU box
with the repo snapshot."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libnsdg.so")
SOURCES = ["nsdg_ctx.hip", "column_step.hip", "transport.hip", "mevp.hip", "mevp_fused.hip", "mevp_fused4.hip", "halo.hip", "rowblock.hip", "forcing.hip"]
HEADERS = ["nsdg_internal.h", "dg_tables.h", "mevp_common.h", "mevp_pipeline.h", "mevp_p2p.h", os.path.join("..", "..", "include", "nsdg.h")]
# -ffp-contract=on: fuse a*b+c only where it is written as one expression (decided in the front end), so
# that the same inlined device function rounds identically in every kernel it is inlined into -- the
# mEVP kernel variants agree bit for bit; the default (fast) let the back end fuse differently per kernel
# and costs only 2 % fewer VALU instructions.
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fno-signed-zeros", "-ffp-contract=on", "-Wall", "-Wno-unused-function"]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libnsdg.so cannot be built")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=True, extra_flags=()):
    extra_flags = tuple(extra_flags) + tuple(os.environ.get("NSDG_EXTRA_FLAGS", "").split())
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(LIBDIR, s.replace(".hip", ".o"))
        cmd = [hipcc()] + FLAGS + list(extra_flags) + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    for s, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            sys.stderr.write(out)
            raise RuntimeError("hipcc failed on %s" % s)
        if verbose and out.strip():
            print(out)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl", "-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


# Diagnostic builds of the same ABI: the sources named in `only` are recompiled with extra flags, everything else is the product's
# objects.  "giveup": every wait of the mEVP pipelines gives up after ONE poll (csrc/mevp_p2p.h) -- the build that lets a test see the
# report channel of a wait that gave up (tests/test_gpu_giveup.py); never loaded by the product.
DIAG = {"giveup": (["-DNSDG_P2P_SPIN_LIMIT=1"], ["mevp_fused4.hip"])}


def diag_lib_path(name):
    return os.path.join(LIBDIR, "diag", name, "libnsdg.so")


def build_diag(name, force=False, verbose=True):
    flags, only = DIAG[name]
    build_lib(verbose=verbose)
    out = diag_lib_path(name)
    deps = [os.path.join(CSRC, s) for s in only + HEADERS] + [os.path.abspath(__file__), LIB]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    objs = []
    for s in SOURCES:
        o = os.path.join(LIBDIR, s.replace(".hip", ".o"))
        if s in only:
            o = os.path.join(os.path.dirname(out), s.replace(".hip", ".o"))
            cmd = [hipcc()] + FLAGS + flags + ["-c", os.path.join(CSRC, s), "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl", "-lpthread"])
    return out


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print("built", LIB)
    for name in DIAG:
        print("built", build_diag(name, force="--force" in sys.argv))
