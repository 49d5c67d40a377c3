"""Single access point to the compute engine used by the host layer.

The product path always resolves to the native HIP engine (`_native.engine()`); it raises if the
shared library or the GPU is missing -- there is no CPU fallback.  Tests of the host logic may
install a stand-in with `set_engine` (e.g. an oracle-backed double for the gloo sharding tests).
"""
import threading

_override = None
_tls = threading.local()


def set_thread_engine(e):
    """Engine used by get_engine() on the calling thread (worker threads of units.UnitPool run on their own HIP stream)."""
    _tls.engine = e


def set_engine(e):
    global _override
    _override = e


_by_stream = {}          # HIP stream handle -> worker engine bound to it (units.UnitPool registers its workers)


def register_stream_engine(stream, e):
    _by_stream[int(stream.cuda_stream)] = e


def get_engine():
    e = getattr(_tls, "engine", None)
    if e is not None:
        return e
    if _override is not None:
        return _override
    if _by_stream:
        # autograd runs the backward of a node on the stream its forward ran on: a node built by a worker thread of the unit pool
        # (differentiable moves) gets that worker's engine back, so that kernels, workspace and torch's allocator agree on the stream
        import torch
        if torch.cuda.is_available():
            w = _by_stream.get(int(torch.cuda.current_stream().cuda_stream))
            if w is not None:
                return w
    import _native
    import config as cfg
    dev = str(getattr(cfg.global_args, "device", "") or "")
    # the engine of the device the host layer was configured for (--GLOBALARGS_device cuda:N), else of the current device
    return _native.engine(dev) if dev.startswith("cuda") else _native.engine()
