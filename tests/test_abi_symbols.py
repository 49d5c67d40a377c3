"""CPU: libctm_hip.so loads and exports every entry point declared in include/ctm_hip.h (no compute calls)."""
import ctypes, os, re
from conftest import REPO, PKG


def _declared():
    src = open(os.path.join(REPO, "include", "ctm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ctm_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    names = _declared()
    for must in ("ctm_create", "ctm_destroy", "ctm_c2x2", "ctm_halves", "ctm_projectors", "ctm_truncated_svd",
                 "ctm_truncated_eigh", "ctm_absorb", "ctm_move_c4v", "ctm_rdm2x2", "ctm_svdvals", "ctm_last_error"):
        assert must in names


def test_library_exports_all_declared_symbols():
    lib_path = os.path.join(PKG, "libctm_hip.so")
    assert os.path.exists(lib_path), "build with python peps-torch_amd/csrc/build.py"
    lib = ctypes.CDLL(lib_path)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing
    lib.ctm_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.ctm_version()


def test_python_binding_covers_the_header():
    import _native
    assert sorted(_native.EXPORTS) == _declared()


def test_no_gpu_means_loud_failure():
    """The product path must not silently fall back to anything when there is no GPU."""
    import torch, pytest, _native, backend
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    backend.set_engine(None)
    with pytest.raises(_native.NativeError):
        backend.get_engine()


def test_fatal_signal_diagnostic_writes_the_native_stack_and_hands_the_signal_on():
    """CTM_ABORT_BACKTRACE=1 (include/ctm_hip.h, ctm_create): on SIGABRT the library writes the native stack of the failing thread to
    fd 2 and then lets the previous owner of the signal act (here Python's faulthandler, then the default action: the process still
    dies of SIGABRT).  Without the variable the library leaves the signals alone.  Runs in child processes; no GPU needed (the handler
    is installed before ctm_create looks for a device)."""
    import signal, subprocess, sys
    code = ("import ctypes, os, faulthandler\n"
            "faulthandler.enable()\n"
            "lib = ctypes.CDLL(%r)\n"
            "h = ctypes.c_void_p()\n"
            "lib.ctm_create(ctypes.byref(h), None, 0)\n"
            "os.abort()\n") % os.path.join(PKG, "libctm_hip.so")
    for armed in (True, False):
        env = dict(os.environ)
        env.pop("CTM_ABORT_BACKTRACE", None)
        if armed:
            env["CTM_ABORT_BACKTRACE"] = "1"
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == -signal.SIGABRT, (armed, r.returncode)
        assert ("ctm_hip: fatal signal" in r.stderr) == armed, r.stderr[-2000:]
        assert "Fatal Python error: Aborted" in r.stderr, r.stderr[-2000:]         # faulthandler still got the signal
        if armed:
            assert "libctm_hip.so" in r.stderr and "abort" in r.stderr
