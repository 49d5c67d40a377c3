"""Build libctm_hip.so (gfx950) in-tree with hipcc.  Usage: python peps-torch_amd/csrc/build.py [--force] [--asan]

--asan: a second library, libctm_hip_asan.so, whose HOST side (dispatchers, arena, Krylov / Jacobi drivers, the C-ABI marshalling) is
compiled with AddressSanitizer (-fsanitize=address -fno-gpu-sanitize: device code unchanged, -O1 -g -fno-omit-frame-pointer).  Run a
test under it with
    CTM_LIB=peps-torch_amd/libctm_hip_asan.so LD_PRELOAD="$(python peps-torch_amd/csrc/build.py --asan-runtime) $(gcc -print-file-name=libstdc++.so.6)" \
    ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=1 python -m pytest tests -m gpu -x -q
(python itself is not instrumented, hence the preload; protect_shadow_gap=0 leaves the address ranges the HSA runtime maps alone)."""
import os, subprocess, sys, shutil

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
SOURCES = ["ctm_runtime.hip", "gemm_f64.hip", "tensor_ops.hip", "jacobi_core.hip", "svd_leading.hip", "eigh.hip", "contract.hip", "layer2.hip", "ctm_ops.hip", "backward.hip", "ctm_move.hip"]
HEADERS = ["ctm_common.h", "contract.h", "jacobi_internal.h", os.path.join("..", "..", "include", "ctm_hip.h")]
LIB = os.path.join(PKG, "libctm_hip.so")
LIB_ASAN = os.path.join(PKG, "libctm_hip_asan.so")


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def asan_runtime():
    """Path of the AddressSanitizer runtime to LD_PRELOAD into the (uninstrumented) python: gcc's libasan.  The runtime of ROCm's own
    clang intercepts the HSA allocation entry points (it is the device-ASAN runtime) and aborts inside torch's bundled HIP runtime
    ("allocator is trying to allocate 0x400000 bytes" in hsa_amd_memory_pool_allocate); gcc's speaks the same __asan_* ABI (v8), has no
    such interceptors, and the library is linked without a runtime of its own."""
    return os.path.realpath(subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip())


def build(force=False, verbose=True, asan=False):
    srcs = [os.path.join(HERE, s) for s in SOURCES]
    deps = srcs + [os.path.join(HERE, h) for h in HEADERS]
    LIB = LIB_ASAN if asan else globals()["LIB"]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    objdir = os.path.join(HERE, "build_asan" if asan else "build")
    os.makedirs(objdir, exist_ok=True)
    cc = _hipcc()
    opt = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address", "-fno-gpu-sanitize"] if asan else ["-O3"]
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s).replace(".hip", ".o"))
        objs.append(o)
        if force or not os.path.exists(o) or any(os.path.getmtime(o) < os.path.getmtime(d) for d in [s] + deps[len(srcs):]):
            cmd = [cc, "--offload-arch=gfx950"] + opt + ["-std=c++17", "-fPIC", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("compile failed: " + " ".join(cmd))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + (["-fsanitize=address", "-fno-gpu-sanitize"] if asan else []) + ["-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    if "--asan-runtime" in sys.argv:
        print(asan_runtime())
    else:
        print(build(force="--force" in sys.argv, asan="--asan" in sys.argv))
