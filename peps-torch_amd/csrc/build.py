"""Build libctm_hip.so (gfx950) in-tree with hipcc.  Usage: python peps-torch_amd/csrc/build.py [--force]"""
import os, subprocess, sys, shutil

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
SOURCES = ["ctm_runtime.hip", "gemm_f64.hip", "tensor_ops.hip", "jacobi.hip", "contract.hip", "layer2.hip", "ctm_ops.hip", "backward.hip"]
HEADERS = ["ctm_common.h", "contract.h", os.path.join("..", "..", "include", "ctm_hip.h")]
LIB = os.path.join(PKG, "libctm_hip.so")


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def build(force=False, verbose=True):
    srcs = [os.path.join(HERE, s) for s in SOURCES]
    deps = srcs + [os.path.join(HERE, h) for h in HEADERS]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    cc = _hipcc()
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s).replace(".hip", ".o"))
        objs.append(o)
        if force or not os.path.exists(o) or any(os.path.getmtime(o) < os.path.getmtime(d) for d in [s] + deps[len(srcs):]):
            cmd = [cc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("compile failed: " + " ".join(cmd))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
