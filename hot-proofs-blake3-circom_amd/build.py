"""Build recipes (hipcc for gfx950, gcc for the oracle and the N-API addon).  In-tree outputs:
  hot-proofs-blake3-circom_amd/libb3wit.so           C-ABI + HIP kernels (the product)
  hot-proofs-blake3-circom_amd/js/b3wit_napi.node    Node.js N-API addon over the C-ABI
  oracle/libb3w_oracle.so                            CPU restatement (test infrastructure)
"""
import os, shutil, subprocess, sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libb3wit.so")
NAPI = os.path.join(PKG, "js", "b3wit_napi.node")
ARCH = "gfx950"


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd, **kw):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, **kw)


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def build_lib(force=False, extra_flags=()):
    """One object per source, compiled side by side (the commit kernels alone take 50 s), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    names = ("b3w_kernels.hip", "b3w_exact.hip", "b3w_plan.hip", "b3w_placement.hip", "b3w_commit.hip", "b3w_r1cs.hip", "b3w_r1cs_walk.hip", "b3w_r1cs_host.cpp", "b3w_hostcomm.cpp",
             "b3w_ctx.cpp", "b3w_bodies.cpp", "b3w_wtns.cpp", "b3w_r1cs_api.cpp", "b3w_commit_api.cpp", "b3w_comm.cpp", "b3w_chain.cpp")
    srcs = [os.path.join(CSRC, f) for f in names]
    hdrs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".inc"))] + \
        [os.path.join(ROOT, "include", "b3wit.h")]
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    if os.environ.get("B3W_BUILD_DIAG") == "1":      # the constraint check's experiment switches (B3W_R1CS_DBG / _STAMPS): never in a product build
        extra_flags = (*extra_flags, "-DB3W_R1CS_DIAG")
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-Wall", "-Wno-unused-function", *extra_flags]
    jobs = []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if force or extra_flags or _newer(obj, [src] + hdrs):
            jobs.append([hipcc(), *flags, "-c", src, "-o", obj])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(_run, jobs))
    objs = [os.path.join(objdir, os.path.basename(src) + ".o") for src in srcs]
    if jobs or _newer(LIB, objs):
        _run([hipcc(), f"--offload-arch={ARCH}", "-fPIC", "-shared", "-o", LIB, *objs, "-ldl", "-lrt", "-lpthread"])
    return LIB


def build_oracle(force=False):
    so = os.path.join(ROOT, "oracle", "libb3w_oracle.so")
    if force or _newer(so, [os.path.join(ROOT, "oracle", "b3w_oracle.c")]):
        _run(["make", "-C", os.path.join(ROOT, "oracle")])
    return so


def build_napi(force=False):
    src = os.path.join(PKG, "js", "b3wit_napi.cc")
    if not os.path.exists(src):
        return None
    inc = "/usr/include/node"
    if not os.path.exists(os.path.join(inc, "node_api.h")):
        print("node_api.h not found; skipping N-API addon", file=sys.stderr)
        return None
    if force or _newer(NAPI, [src, os.path.join(ROOT, "include", "b3wit.h")]):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DNODE_GYP_MODULE_NAME=b3wit_napi", "-I", inc,
              "-I", os.path.join(ROOT, "include"), "-o", NAPI, src, "-ldl"])
    return NAPI


def build_all(force=False):
    build_lib(force)
    build_oracle(force)
    build_napi(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
