"""ctypes binding of libctm_hip.so (the C-ABI in include/ctm_hip.h) for torch device tensors.

torch is plumbing only: it owns device memory and the HIP stream; every contraction,
decomposition and reduction of the hot path runs inside the native library.  There is NO
fallback: if the shared library is missing or a call fails, this module raises.
"""
import ctypes as C
import os
import threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # see config.py: one hardware queue per worker stream
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# CTM_LIB: another build of the same library (csrc/build.py --asan: host side under AddressSanitizer); never anything but libctm_hip
_LIBPATH = os.environ.get("CTM_LIB") or os.path.join(_HERE, "libctm_hip.so")

CTM_OK = 0
CTM_ERR_NOMEM = 6
_ERRNAMES = {1: "bad argument", 2: "shape mismatch", 3: "no convergence", 4: "HIP error", 5: "unsupported", 6: "out of memory",
             7: "context busy (used by two threads at once)"}
LU, RU, RD, LD = 0, 1, 2, 3
UP, LEFT, DOWN, RIGHT = 0, 1, 2, 3
DIR_INDEX = {(0, -1): UP, (-1, 0): LEFT, (0, 1): DOWN, (1, 0): RIGHT}
# leg of the anchor site a[s,u,l,d,r] that lies on the cut truncated by a move (UP: l, LEFT: d, DOWN: r, RIGHT: u)
CUT_LEG = {UP: 2, LEFT: 3, DOWN: 4, RIGHT: 1}


class NativeError(RuntimeError):
    status = None          # the library's status code when the error came from a C entry (include/ctm_hip.h: CTM_ERR_*)


class TruncCfg(C.Structure):
    _fields_ = [("svd_reltol", C.c_double), ("eps_multiplet", C.c_double), ("multiplet_abstol", C.c_double),
                ("keep_multiplets", C.c_int), ("fix_signs", C.c_int)]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_longlong)      # include/ctm_hip.h: ctm_allgather_fn


class _NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


class MoveUnit(C.Structure):
    """include/ctm_hip.h: ctm_move_unit (one site of a whole-move call)."""
    _fields_ = [("proj", C.c_void_p * 16), ("proj_adims", C.c_int * 20), ("basis", C.c_void_p), ("corner_buf", C.c_void_p * 4),
                ("corner_valid", C.c_int * 4), ("use_corner_cache", C.c_int), ("absorb", C.c_void_p * 6), ("absorb_adims", C.c_int * 5),
                ("nb", C.c_int), ("n_rows", C.c_longlong), ("P", C.c_void_p), ("Pt", C.c_void_p), ("S", C.c_void_p),
                ("nC1", C.c_void_p), ("nC2", C.c_void_p), ("nT", C.c_void_p), ("ncol", C.c_int)]


_lib = None

_SIGS = {
    "ctm_create": [C.POINTER(C.c_void_p), C.c_void_p, C.c_int],
    "ctm_destroy": [C.c_void_p],
    "ctm_sync": [C.c_void_p],
    "ctm_trim": [C.c_void_p],
    "ctm_set_option": [C.c_void_p, C.c_char_p, C.c_double],
    "ctm_projectors_rect": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(TruncCfg), C.c_void_p, C.c_void_p, C.c_void_p],
    "ctm_set_comm": [C.c_void_p, C.c_void_p, C.c_int, C.c_int],
    "ctm_set_comm_ops": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int],
    "ctm_get_stat": [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)],
    "ctm_timers": [C.c_void_p, C.POINTER(C.c_double), C.c_int],
    "ctm_gemm_intervals": [C.c_void_p, C.POINTER(C.c_double), C.c_longlong, C.POINTER(C.c_longlong)],
    "ctm_gemm": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_longlong,
                 C.c_void_p, C.c_longlong, C.c_double, C.c_void_p, C.c_longlong],
    "ctm_permute": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_int)],
    "ctm_normalize_inf": [C.c_void_p, C.c_void_p, C.c_longlong],
    "ctm_einsum": [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.POINTER(C.c_int), C.c_void_p],
    "ctm_truncated_svd": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(TruncCfg), C.c_void_p, C.c_void_p, C.c_void_p],
    "ctm_truncated_svd_ws": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(TruncCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "ctm_truncated_eigh": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(TruncCfg), C.c_void_p, C.c_void_p],
    "ctm_truncated_eigh_ws": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(TruncCfg), C.c_void_p, C.c_void_p, C.c_void_p],
    "ctm_svd_symeig": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(TruncCfg), C.c_void_p, C.c_void_p, C.c_void_p],
    "ctm_svdvals": [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p],
    "ctm_svd_backward": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                         C.c_double, C.c_void_p],
    "ctm_eigh_backward": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p],
    "ctm_c2x2": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p],
    "ctm_halves": [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p],
    "ctm_projectors": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(TruncCfg), C.c_void_p, C.c_void_p, C.c_void_p],
    "ctm_projectors_4x4": [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.POINTER(TruncCfg), C.c_void_p, C.c_void_p, C.c_void_p],
    "ctm_projectors_4x4_ws": [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.POINTER(TruncCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "ctm_projectors_4x4_cc": [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.POINTER(TruncCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.POINTER(C.c_void_p), C.POINTER(C.c_int)],
    "ctm_absorb": [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    "ctm_absorb_x": [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    "ctm_c2x2_c4v": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "ctm_move_c4v": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(TruncCfg),
                     C.c_void_p, C.c_void_p, C.c_void_p],
    "ctm_move_c4v_ws": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(TruncCfg),
                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "ctm_move_c4v_x": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(TruncCfg), C.c_int,
                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "ctm_rdm2x2": [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.c_void_p],
    "ctm_rdm2x2_part": [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p],
    "ctm_rdm1x1": [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.c_void_p],
    "ctm_rdm2x1": [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.c_void_p],
    "ctm_rdm1x2": [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.c_void_p],
    "ctm_rdm_c4v": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "ctm_init_piece": [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_void_p],
    "ctm_move": [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.POINTER(MoveUnit), C.c_int, C.POINTER(TruncCfg), C.c_int, C.c_int],
}
EXPORTS = sorted(list(_SIGS) + ["ctm_last_error", "ctm_version"])


def load_library(path=None):
    """Load libctm_hip.so; raises NativeError when it is not there (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or _LIBPATH
    if not os.path.exists(path):
        raise NativeError(f"{path} not found: build it with `python peps-torch_amd/csrc/build.py` "
                          "(the engine has no CPU fallback)")
    lib = C.CDLL(path)
    for name, args in _SIGS.items():
        f = getattr(lib, name)
        f.argtypes = args
        f.restype = C.c_int
    lib.ctm_last_error.argtypes = [C.c_void_p]
    lib.ctm_last_error.restype = C.c_char_p
    lib.ctm_version.argtypes = []
    lib.ctm_version.restype = C.c_char_p
    _lib = lib
    return lib


def _ptr(t):
    return C.c_void_p(t.data_ptr())


_DTYPES = {torch.float64: 0, torch.complex128: 1}     # CTM_F64, CTM_C128


def _chk_t(t, name="tensor", dtype=None):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype in _DTYPES and (dtype is None or t.dtype == dtype)):
        raise NativeError(f"{name}: expected a {dtype or 'float64/complex128'} CUDA(HIP) tensor, got {type(t)} "
                          f"{getattr(t, 'dtype', None)} {getattr(t, 'device', None)}")
    return _plain(t)


def _plain(t):
    """The tensor as the library reads it: contiguous memory holding the VALUES.  torch represents `x.conj()` (and `-x` in some
    autograd paths) lazily as a flag on the same memory -- e.g. the gradient that flows back through `V.conj().transpose(-2, -1)` --
    so the flags are materialised before a pointer is taken."""
    if t.is_complex() and t.is_conj():
        t = t.resolve_conj()
    if t.is_neg():
        t = t.resolve_neg()
    return t if t.is_contiguous() else t.contiguous()


class Engine:
    """Native contexts (one per dtype: float64, complex128) bound to a torch device and its current stream.

    Every call binds the context matching the dtype of its tensor arguments (`_bind`); a complex128 tensor is handed
    to the library as torch stores it (interleaved re,im)."""

    def __init__(self, device=None):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise NativeError("no HIP device visible: the CTM engine runs only on the GPU")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._handles = {}
        self._options = {}
        self._tls = threading.local()          # the context selected by _bind is per CALLING THREAD (h, dtype below)
        self._create_lock = threading.Lock()
        self.workers = []          # engines spawned for concurrent units (units.UnitPool): options/stats fan out to them
        self.default_cfg = TruncCfg(1e-8, 1e-8, 1e-14, 1, 1)
        self._bind_dtype(torch.float64)

    # The handle / dtype a call works with.  Thread-local: autograd replays backward nodes on its own device thread while another
    # thread may be inside a call of the same engine object with the other dtype; as plain instance attributes, a `_bind` of one
    # thread changed the context the other was about to pass to the library (a complex128 context handed
    # float64 buffers over-reads them).  Concurrent use of ONE native context is still refused by the library (CTM_ERR_BUSY).
    @property
    def h(self):
        return self._tls.h

    @h.setter
    def h(self, v):
        self._tls.h = v

    @property
    def dtype(self):
        return self._tls.dtype

    @dtype.setter
    def dtype(self, v):
        self._tls.dtype = v

    def spawn_worker(self):
        """A sibling engine on the CALLER's current stream (own native contexts and arenas), inheriting the options."""
        w = Engine(self.device)
        for k, v in self._options.items():
            w.set_option(k, v)
        self.workers.append(w)
        return w

    def _bind_dtype(self, dtype):
        if dtype not in _DTYPES:
            raise NativeError(f"unsupported dtype {dtype}: the engine computes in float64 or complex128")
        if dtype not in self._handles:
            with self._create_lock:
                if dtype not in self._handles:
                    with torch.cuda.device(self.device):
                        # every context of an engine runs on the stream the engine was created on (the float64 context is made in
                        # __init__): a complex128 context made lazily from inside some other stream scope must not bind that stream
                        if getattr(self, "stream", None) is None:
                            self.stream = torch.cuda.current_stream(self.device)
                        h = C.c_void_p()
                        st = self.lib.ctm_create(C.byref(h), C.c_void_p(self.stream.cuda_stream), _DTYPES[dtype])
                        if st != CTM_OK:
                            raise NativeError(f"ctm_create failed: {_ERRNAMES.get(st, st)}")
                    for k, v in self._options.items():
                        self.lib.ctm_set_option(h, k.encode(), float(v))
                    self._handles[dtype] = h
        self.h = self._handles[dtype]
        self.dtype = dtype

    def _bind(self, *tensors):
        """Select the native context for the dtype of the tensor arguments and validate them."""
        dt = tensors[0].dtype if isinstance(tensors[0], torch.Tensor) else None
        # the native contexts (arena, stream, kernels) live on self.device: make it the calling thread's current HIP device and
        # refuse tensors of another GPU instead of launching on device-A pointers from device B
        if torch.cuda.current_device() != self.device.index:
            torch.cuda.set_device(self.device)
        for t in tensors:
            if isinstance(t, torch.Tensor) and t.is_cuda and t.device != self.device:
                raise NativeError(f"tensor on {t.device} passed to the engine of {self.device}: use _native.engine(tensor.device)")
        self._bind_dtype(dt)
        out = [_chk_t(t, "tensor", dt) for t in tensors]
        return out[0] if len(out) == 1 else out

    def close(self):
        """Destroy the native contexts of this engine and of its workers (idempotent)."""
        for w in getattr(self, "workers", []):
            w.close()
        self.workers = []
        for h in getattr(self, "_handles", {}).values():
            try:
                self.lib.ctm_destroy(h)
            except Exception:
                pass
        self._handles = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st, what):
        if st != CTM_OK:
            msg = self.lib.ctm_last_error(self.h)
            err = NativeError(f"{what}: {_ERRNAMES.get(st, st)}: {msg.decode() if msg else ''}")
            err.status = st                  # include/ctm_hip.h status code (CTM_ERR_NOMEM = 6: callers may retry with less in flight)
            raise err

    def empty(self, *shape):
        return torch.empty(shape, dtype=self.dtype, device=self.device)

    def empty_real(self, *shape):
        return torch.empty(shape, dtype=torch.float64, device=self.device)

    # ---- options / stats (applied to / summed over the contexts of both dtypes) ----------------
    def set_option(self, key, value):
        self._options[key] = value
        for w in self.workers:
            w.set_option(key, value)
        for h in self._handles.values():
            st = self.lib.ctm_set_option(h, key.encode(), float(value))
            if st != CTM_OK:
                self.h = h
                self._ck(st, "set_option")

    def set_comm(self, comm, rank, nranks):
        """Rank group that shares this engine's units (include/ctm_hip.h: ctm_set_comm; a one-rank group in this build)."""
        for h in self._handles.values():
            st = self.lib.ctm_set_comm(h, C.c_void_p(comm or 0), int(rank), int(nranks))
            if st != CTM_OK:
                self.h = h
                self._ck(st, "set_comm")

    # ---- a raw RCCL communicator for a rank group (the transport ctm_set_comm takes: ncclAllGather on the library's own stream) ----------
    _rccl = None

    @classmethod
    def _librccl(cls):
        if cls._rccl is None:
            import glob
            cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")) + ["librccl.so", "librccl.so.1"]
            for c in cands:
                try:
                    cls._rccl = C.CDLL(c)
                    break
                except OSError:
                    pass
            if cls._rccl is None:
                raise NativeError("librccl.so is not loadable in this process")
            cls._rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclUniqueId, C.c_int]
            cls._rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        return cls._rccl

    def attach_rccl_group(self, members):
        """ncclCommInitRank over `members` (global ranks of the default torch.distributed group; the unique id travels from the first member
        through torch.distributed) and ctm_set_comm: the corner passes of this engine's float64 units are then split over the group and
        all-gathered with ncclAllGather on the engine's stream.  One communicator per group, kept for the life of the engine.
        Executed in the build loop with ONE member only (tools/check_rccl_world1.py); opt-in for pairs: CTM_GROUP_TRANSPORT=rccl."""
        import torch.distributed as dist
        import parallel
        key = tuple(members)
        cache = self.__dict__.setdefault("_rccl_comms", {})
        rank_in = list(members).index(dist.get_rank())
        if key not in cache:
            lib = self._librccl()
            uid = _NcclUniqueId()
            if rank_in == 0 and lib.ncclGetUniqueId(C.byref(uid)) != 0:
                raise NativeError("ncclGetUniqueId failed")
            box = [C.string_at(C.byref(uid), 128) if rank_in == 0 else None]      # (all 128 bytes: a c_char array field reads as a C string)
            if dist.is_initialized():
                grp = parallel._process_group(list(members)) if len(members) < dist.get_world_size() else None
                dist.broadcast_object_list(box, src=members[0], group=grp)
            C.memmove(C.byref(uid), box[0], 128)
            comm = C.c_void_p()
            with torch.cuda.device(self.device):
                if lib.ncclCommInitRank(C.byref(comm), len(members), uid, rank_in) != 0:
                    raise NativeError("ncclCommInitRank failed")
            cache[key] = comm
        self.set_comm(cache[key].value, rank_in, len(members))

    def detach_group(self):
        self.set_comm(None, 0, 1)

    def set_group(self, members, capacity_doubles=0):
        """Rank group (global ranks of the default torch.distributed group, this rank among them; at most two) that shares the units of this
        engine's float64 context: every corner pass of the implicit operator computes this rank's column block and the group all-gathers
        the blocks (include/ctm_hip.h: ctm_set_comm_ops -- the host-driven form: torch.distributed on whatever backend the job runs,
        gloo in the build loop's tests, RCCL on a node).  capacity_doubles: rows x columns / len(members) of the largest pass.
        members None or one rank: detach.  Every rank of the group must make the same engine calls while it is attached."""
        import torch.distributed as dist
        h = self._handles[torch.float64]
        if os.environ.get("CTM_GROUP_TRANSPORT") == "rccl":           # (opt-in: never executed with two ranks -- DESIGN.md section 6)
            if members and len(members) >= 2:
                return self.attach_rccl_group(list(members))
            return self.detach_group()
        if not members or len(members) < 2:
            st = self.lib.ctm_set_comm_ops(h, None, None, None, None, 0, 0, 1)
            self._group = None
            if st != CTM_OK:
                self.h = h; self._ck(st, "set_comm_ops")
            return
        import parallel
        g = len(members)
        rank_in = list(members).index(dist.get_rank())
        grp = parallel._process_group(list(members))
        cap = int(capacity_doubles)
        keep = getattr(self, "_group_bufs", None)
        if keep is None or keep[0].numel() < cap:
            keep = self._group_bufs = (torch.empty(cap, dtype=torch.float64, device=self.device), torch.empty(g * cap, dtype=torch.float64, device=self.device))
        send, recv = keep

        def allgather(user, count):
            try:
                dist.all_gather_into_tensor(recv[:g * count], send[:count], group=grp)
                torch.cuda.synchronize(self.device)
                return 0
            except Exception as e:                    # (a Python exception must not unwind through the C frames)
                self._group_error = repr(e)
                return 1
        cb = ALLGATHER_FN(allgather)
        self._group = (cb, grp, tuple(members))        # keeps the callback alive while the library holds it
        st = self.lib.ctm_set_comm_ops(h, C.cast(cb, C.c_void_p), None, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), send.numel(), rank_in, g)
        if st != CTM_OK:
            self.h = h; self._ck(st, "set_comm_ops")

    def stat(self, key):
        tot = sum(w.stat(key) for w in self.workers)
        for h in self._handles.values():
            v = C.c_double()
            st = self.lib.ctm_get_stat(h, key.encode(), C.byref(v))
            if st != CTM_OK:
                self.h = h
                self._ck(st, "get_stat")
            tot += v.value
        return tot

    def own_stat(self, key):
        """stat() of this engine's contexts only (no workers)."""
        tot = 0.0
        for h in self._handles.values():
            v = C.c_double()
            if self.lib.ctm_get_stat(h, key.encode(), C.byref(v)) == CTM_OK:
                tot += v.value
        return tot

    def timers(self, reset=False):
        names = ["corners", "halves", "svd", "proj", "absorb", "norm", "rdm", "eig"]
        tot = dict.fromkeys(names, 0.0)
        for w in self.workers:
            for n, v in w.timers(reset).items():
                tot[n] += v
        for h in self._handles.values():
            buf = (C.c_double * 8)()
            self.lib.ctm_timers(h, buf, int(reset))
            for n, v in zip(names, list(buf)):
                tot[n] += v
        return tot

    def trim(self, workers_only=False):
        """Give the workspace arenas back to the device (they regrow on demand): all worker contexts, and this engine's."""
        for w in self.workers:
            w.trim()
        if not workers_only:
            for h in self._handles.values():
                self.lib.ctm_trim(h)

    def trim_own(self):
        """Give back the workspace arenas of THIS engine's contexts only (the units of a move are running on the workers)."""
        for h in self._handles.values():
            self.lib.ctm_trim(h)

    def gemm_intervals(self):
        """(kind, start_ms, end_ms, flops) of every GEMM launch timed since "gemm_timing" was switched on, over all
        contexts of this engine and of its workers (one process-wide clock)."""
        out = []
        for w in self.workers:
            out += w.gemm_intervals()
        for h in self._handles.values():
            cnt = C.c_longlong()
            self.lib.ctm_gemm_intervals(h, None, 0, C.byref(cnt))
            if cnt.value:
                buf = (C.c_double * (4 * cnt.value))()
                self.lib.ctm_gemm_intervals(h, buf, 4 * cnt.value, C.byref(cnt))
                v = list(buf)
                out += [tuple(v[i:i + 4]) for i in range(0, len(v), 4)]
        return out

    def sync(self):
        for w in self.workers:
            w.sync()
        for h in self._handles.values():
            self.lib.ctm_sync(h)

    def cfg(self, svd_reltol=1e-8, eps_multiplet=1e-8, multiplet_abstol=1e-14, keep_multiplets=True, fix_signs=True):
        return TruncCfg(svd_reltol, eps_multiplet, multiplet_abstol, int(keep_multiplets), int(fix_signs))

    # ---- primitives ---------------------------------------------------------------------------
    def gemm(self, A, B, transA=False, transB=False, alpha=1.0, beta=0.0, out=None):
        """alpha op(A) op(B) (+ beta out).  `out` (float64 only): an M x ldc row-major tensor, ldc >= N, updated in its first N columns."""
        A, B = self._bind(A, B)          # complex128: trans = 0/False 'N', 1/True 'T', 2 'C' (conjugate transpose)
        M, K = (A.shape[1], A.shape[0]) if transA else A.shape
        K2, N = (B.shape[1], B.shape[0]) if transB else B.shape
        if K != K2:
            raise NativeError("gemm: inner dimensions differ")
        if out is None:
            if beta != 0.0:
                raise NativeError("gemm: beta != 0 needs `out`")
            out = self.empty(M, N)
        else:
            if not (isinstance(out, torch.Tensor) and out.is_cuda and out.dtype == A.dtype and out.is_contiguous()) or (out.is_complex() and out.is_conj()) or out.is_neg():
                raise NativeError("gemm: `out` must be a contiguous device tensor of the operands' dtype (a column slice would be written into a temporary copy)")
            if out.dim() != 2 or out.shape[0] != M or out.shape[1] < N:
                raise NativeError("gemm: `out` must be M x ldc with ldc >= N")
            if out.is_complex() and (out.shape[1] != N or beta != 0.0):
                raise NativeError("gemm: complex128 `out` must be exactly M x N with beta = 0")
        self._ck(self.lib.ctm_gemm(self.h, int(transA), int(transB), M, N, K, alpha, _ptr(A), A.shape[1], _ptr(B),
                                   B.shape[1], beta, _ptr(out), out.shape[1]), "gemm")
        return out

    def permute(self, x, perm):
        x = self._bind(x)
        nd = x.dim()
        out = self.empty(*[x.shape[p] for p in perm])
        dims = (C.c_longlong * nd)(*x.shape)
        pm = (C.c_int * nd)(*perm)
        self._ck(self.lib.ctm_permute(self.h, _ptr(x), _ptr(out), nd, dims, pm), "permute")
        return out

    def einsum(self, expr, *tensors, conj=()):
        """Native contraction of "i0,i1,...->o" (left to right); conj: indices of the operands read conjugated."""
        ts = self._bind(*tensors)
        ts = ts if isinstance(ts, list) else [ts]
        lhs, o = expr.split("->")
        ins = lhs.split(",")
        ext = {}
        for idx, t in zip(ins, ts):
            if len(idx) != t.dim():
                raise NativeError(f"einsum: operand '{idx}' has rank {t.dim()}")
            for ch, n in zip(idx, t.shape):
                ext[ch] = n
        out = self.empty(*[ext[ch] for ch in o])
        arr = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        nd = (C.c_int * len(ts))(*[t.dim() for t in ts])
        flat = [n for t in ts for n in t.shape]
        dims = (C.c_longlong * len(flat))(*flat)
        cj = (C.c_int * len(ts))(*[1 if i in conj else 0 for i in range(len(ts))])
        self._ck(self.lib.ctm_einsum(self.h, expr.encode(), len(ts), arr, nd, dims, cj, _ptr(out)), "einsum")
        return out

    def normalize_inf_(self, x):
        self._bind(x)
        self._ck(self.lib.ctm_normalize_inf(self.h, _ptr(x), x.numel()), "normalize_inf")
        return x

    # ---- truncation -----------------------------------------------------------------------------
    def truncated_svd(self, M, chi, cfg=None, basis=None):
        """basis: optional warm-start workspace from warm_basis(chi, n, dtype), updated in place (sequence of nearby matrices)."""
        M = self._bind(M)
        n = M.shape[0]
        if M.dim() != 2 or M.shape[1] != n:
            raise NativeError("truncated_svd: square matrices only on this path")
        kc = min(chi, n)
        U, S, V = self.empty(n, kc), self.empty_real(kc), self.empty(n, kc)
        cfg = cfg or self.default_cfg
        if basis is not None:
            if not (basis.is_cuda and basis.dtype == torch.float64 and basis.is_contiguous()
                    and tuple(basis.shape) == (self.warm_rows(chi, n, M.dtype)[0], n)):
                raise NativeError("truncated_svd: basis must come from warm_basis(chi, n, dtype)")
        self._ck(self.lib.ctm_truncated_svd_ws(self.h, _ptr(M), n, chi, C.byref(cfg), _ptr(U), _ptr(S), _ptr(V),
                                               _ptr(basis) if basis is not None else None), "truncated_svd")
        return U, S, V

    def truncated_eigh(self, A, chi, cfg=None, basis=None):
        """basis: optional workspace from warm_basis_c4v(chi, n), updated in place (sequence of nearby symmetric matrices)."""
        A = self._bind(A)
        n = A.shape[0]
        kc = min(chi, n)
        D, U = self.empty_real(kc), self.empty(n, kc)
        cfg = cfg or self.cfg(eps_multiplet=1e-12)
        if basis is not None:
            k = chi + 1 if chi < n else n
            if not (basis.is_cuda and basis.dtype == torch.float64 and basis.is_contiguous()
                    and tuple(basis.shape) == ((2 if A.dtype.is_complex else 1) * min(n, k + 8) + 1, n)):
                raise NativeError("truncated_eigh: basis must come from warm_basis_c4v(chi, n, dtype)")
            self._ck(self.lib.ctm_truncated_eigh_ws(self.h, _ptr(A), n, chi, C.byref(cfg), _ptr(D), _ptr(U), _ptr(basis)), "truncated_eigh")
        else:
            self._ck(self.lib.ctm_truncated_eigh(self.h, _ptr(A), n, chi, C.byref(cfg), _ptr(D), _ptr(U)), "truncated_eigh")
        return D, U

    def svd_symeig(self, A, chi=None, cfg=None):
        """SVD of a real symmetric matrix from its eigendecomposition: U, S = |D|, V = U sign(D), ordered by |D| descending."""
        A = self._bind(A)
        n = A.shape[0]
        chi = n if chi is None else chi
        kc = min(chi, n)
        U, S, V = self.empty(n, kc), self.empty_real(kc), self.empty(n, kc)
        cfg = cfg or self.cfg(eps_multiplet=1e-12, keep_multiplets=False)
        self._ck(self.lib.ctm_svd_symeig(self.h, _ptr(A), n, chi, C.byref(cfg), _ptr(U), _ptr(S), _ptr(V)), "svd_symeig")
        return U, S, V

    def svd_backward(self, U, S, V, gU=None, gS=None, gV=None, eps=1.0e-12):
        """dA of A = U diag(S) V^H given the gradients on U, S, V (reference SVDGESDD.backward); thin factors allowed."""
        U, V = self._bind(U, V)
        m, k = U.shape
        n = V.shape[0]
        S = _plain(S)
        g = [None if t is None else _plain(t.to(U.dtype) if i != 1 else t.to(torch.float64)) for i, t in enumerate((gU, gS, gV))]
        dA = self.empty(m, n)
        opt = lambda t: _ptr(t) if t is not None else None
        self._ck(self.lib.ctm_svd_backward(self.h, _ptr(U), _ptr(S), _ptr(V), opt(g[0]), opt(g[1]), opt(g[2]), m, n, k, float(eps), _ptr(dA)),
                 "svd_backward")
        return dA

    def eigh_backward(self, D, U, gD=None, gU=None, reg=1.0e-12):
        """dA of A = U diag(D) U^H given the gradients on D and U (reference SYMEIG.backward)."""
        U = self._bind(U)
        n, k = U.shape
        D = _plain(D)
        gD = None if gD is None else _plain(gD.to(torch.float64))
        gU = None if gU is None else _plain(gU.to(U.dtype))
        dA = self.empty(n, n)
        opt = lambda t: _ptr(t) if t is not None else None
        self._ck(self.lib.ctm_eigh_backward(self.h, _ptr(D), _ptr(U), opt(gD), opt(gU), n, k, float(reg), _ptr(dA)), "eigh_backward")
        return dA

    def svdvals(self, M):
        M = self._bind(M)
        n = M.shape[0]
        S = self.empty_real(n)
        self._ck(self.lib.ctm_svdvals(self.h, _ptr(M), n, _ptr(S)), "svdvals")
        return S

    # ---- generic move units -----------------------------------------------------------------------
    @staticmethod
    def _adims(a):
        return (C.c_int * 5)(*a.shape)

    def c2x2(self, corner, C_, T1, T2, a, open_=False):
        C_, T1, T2, a = self._bind(C_, T1, T2, a)
        chi = C_.shape[0]
        ad = a.shape
        leg0 = (3, 2, 1, 1)[corner]; leg1 = (4, 3, 2, 4)[corner]
        n0, n1 = chi * ad[leg0] ** 2, chi * ad[leg1] ** 2
        out = self.empty(n0, n1, ad[0], ad[0]) if open_ else self.empty(n0, n1)
        self._ck(self.lib.ctm_c2x2(self.h, corner, int(open_), _ptr(C_), _ptr(T1), _ptr(T2), _ptr(a), chi,
                                   self._adims(a), _ptr(out)), "c2x2")
        return out

    def _pack16(self, tensors):
        ts = self._bind(*tensors)
        arr = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        ad = []
        for i in range(3, len(ts), 4):
            ad += list(ts[i].shape)
        return ts, arr, (C.c_int * len(ad))(*ad)

    def halves(self, direction, tensors16):
        """tensors16: (C,T1,T2,a) of corner A of R, corner B of R, corner A of Rt, corner B of Rt."""
        ts, arr, ad = self._pack16(tensors16)
        chi = ts[0].shape[0]
        d = DIR_INDEX[direction] if isinstance(direction, tuple) else direction
        # R: (truncated bond at the site of its first corner) x (the same leg at the site of its second corner); Rt: the opposite legs of
        # its two sites.  Square unless bond dimensions differ along the cut (rectangular halves: projectors(R, Rt) handles both)
        leg = CUT_LEG[d]; opp = {1: 3, 2: 4, 3: 1, 4: 2}[leg]
        shR = (chi * ts[3].shape[leg] ** 2, chi * ts[7].shape[leg] ** 2)
        shRt = (chi * ts[11].shape[opp] ** 2, chi * ts[15].shape[opp] ** 2)
        R, Rt = self.empty(*shR), self.empty(*shRt)
        self._ck(self.lib.ctm_halves(self.h, d, arr, chi, ad, _ptr(R), _ptr(Rt)), "halves")
        return R, Rt

    def projectors(self, R, Rt, chi, cfg=None, return_S=False):
        R, Rt = self._bind(R, Rt)
        if R.shape != Rt.shape or R.dim() != 2:
            raise AssertionError("R and Rt must be matrices of equal shape")     # ctm_projectors.py:209
        a, b = R.shape                         # rectangular halves (bond dimensions that differ along one cut): M = R^T Rt is b x b
        kc = min(chi, b)
        P, Pt, S = self.empty(a, kc), self.empty(a, kc), self.empty_real(kc)
        cfg = cfg or self.default_cfg
        self._ck(self.lib.ctm_projectors_rect(self.h, _ptr(R), _ptr(Rt), a, b, chi, C.byref(cfg), _ptr(P), _ptr(Pt), _ptr(S)), "projectors")
        return (P, Pt, S) if return_S else (P, Pt)

    @staticmethod
    def warm_rows(chi, n, dtype):
        """Rows (of n doubles) of a warm-start workspace: the k = min(chi + 1, n) basis rows (two planes for complex128), the header row, and
        -- for the sizes the block Krylov solvers take (n >= 256, k >= 48, k < n) -- the Ritz region behind it: 16 words + the accumulated
        rotations of the unit's last Ritz extraction (m x m, m up to ~4.5 k rows of Krylov basis; twice that for complex128), from which
        the next extraction starts (csrc/svd_leading.hip: HDR_RITZ_CAP).  Returns (rows, ritz_doubles)."""
        k = chi + 1 if chi < n else n
        cz = 2 if dtype.is_complex else 1
        ritz_rows = 0
        if k < n and n >= 256 and k >= 48:
            mcap = min((n // 2) // 64 * 64 + 64, (int(4.5 * k) + 127) // 64 * 64)     # (+ the dense solver's padding to an even panel count)
            ritz_rows = -(-(16 + cz * mcap * mcap) // n)
        return cz * k + 1 + ritz_rows, ritz_rows * n

    def warm_basis(self, chi, n, dtype):
        """Zero-filled warm-start workspace for projectors_4x4(..., basis=) / truncated_svd(..., basis=) of one (direction, site) unit
        (layout: warm_rows)."""
        rows, ritz = self.warm_rows(chi, n, dtype)
        b = torch.zeros(rows, n, dtype=torch.float64, device=self.device)
        if ritz:
            k = chi + 1 if chi < n else n
            b[(2 if dtype.is_complex else 1) * k, 10] = float(ritz)      # header word HDR_RITZ_CAP: doubles behind the header row
        return b

    def corner_numel(self, corner, C_, a):
        """Doubles of an opaque corner buffer for projectors_4x4(corners=...): n0 * n1 (twice that for complex128)."""
        chi = C_.shape[0]
        leg0 = (3, 2, 1, 1)[corner]; leg1 = (4, 3, 2, 4)[corner]
        return chi * a.shape[leg0] ** 2 * chi * a.shape[leg1] ** 2 * (2 if C_.dtype.is_complex else 1)

    def projectors_4x4(self, direction, tensors16, chi, cfg=None, return_S=False, basis=None, corners=None):
        """Fused corners -> implicit R^T Rt -> leading-chi triplets -> P, Pt (never forms the n x n halves).
        basis: optional warm-start workspace (see warm_basis), updated in place.
        corners: optional list of four (buffer, valid) pairs -- float64 device buffers of corner_numel() doubles kept by the
        caller; valid=True: the buffer holds that enlarged corner from an earlier call with the same input tensors."""
        ts, arr, ad = self._pack16(tensors16)
        d = DIR_INDEX[direction] if isinstance(direction, tuple) else direction
        chi_env = ts[0].shape[0]
        n = chi_env * ts[3].shape[CUT_LEG[d]] ** 2
        kc = min(chi, n)
        P, Pt, S = self.empty(n, kc), self.empty(n, kc), self.empty_real(kc)
        cfg = cfg or self.default_cfg
        if basis is not None:
            if not (basis.is_cuda and basis.dtype == torch.float64 and basis.is_contiguous()
                    and tuple(basis.shape) == (self.warm_rows(chi, n, ts[0].dtype)[0], n)):
                raise NativeError("projectors_4x4: basis must come from warm_basis(chi, n, dtype)")
        cb = cv = None
        if corners is not None:
            for buf, _ in corners:
                if buf is not None and not (buf.is_cuda and buf.dtype == torch.float64 and buf.is_contiguous()):
                    raise NativeError("projectors_4x4: corner buffers must be contiguous float64 device tensors")
            cb = (C.c_void_p * 4)(*[(_ptr(buf) if buf is not None else None) for buf, _ in corners])
            cv = (C.c_int * 4)(*[int(bool(v)) for _, v in corners])
        st = self.lib.ctm_projectors_4x4_cc(self.h, d, arr, chi, ad, C.byref(cfg), _ptr(P), _ptr(Pt), _ptr(S),
                                            _ptr(basis) if basis is not None else None, cb, cv)
        if st == 5:
            # the entry's shape check (before any kernel): the reference only asserts R.shape == Rt.shape (ctm_projectors.py:209); this
            # engine truncates SQUARE halves -- the two bonds a cut crosses must have the same dimension
            msg = self.lib.ctm_last_error(self.h)
            raise ValueError("ctm_get_projectors_4x4: " + (msg.decode() if msg else "unsupported shapes"))
        self._ck(st, "projectors_4x4")
        return (P, Pt, S) if return_S else (P, Pt)

    # axis of the NEW (truncated) bond in nC1 / nC2 per direction (output index order of the absorb contractions)
    NEW_AXIS = {UP: (0, 1), LEFT: (0, 0), DOWN: (1, 1), RIGHT: (0, 1)}

    def absorb(self, direction, tensors10, normalize=True):
        """nC1, nC2, nT of one site.  The environment tensors may be smaller (chi_in) than the truncated bond (chi_out =
        projector columns): nC1 / nC2 then come back as chi_out x chi_in (or transposed, see NEW_AXIS)."""
        ts = self._bind(*tensors10)
        arr = (C.c_void_p * 10)(*[t.data_ptr() for t in ts])
        A = ts[5]
        chi_in = ts[0].shape[0]
        chi_out = ts[6].shape[1]
        d = DIR_INDEX[direction] if isinstance(direction, tuple) else direction
        out_leg = (3, 4, 1, 2)[d]
        D2 = A.shape[out_leg] ** 2
        shapes = {UP: (chi_out, D2, chi_out), LEFT: (chi_out, chi_out, D2), DOWN: (D2, chi_out, chi_out), RIGHT: (chi_out, D2, chi_out)}
        cs = lambda new_axis: (chi_out, chi_in) if new_axis == 0 else (chi_in, chi_out)
        n1, n2 = self.NEW_AXIS[d]
        nC1, nC2, nT = self.empty(*cs(n1)), self.empty(*cs(n2)), self.empty(*shapes[d])
        self._ck(self.lib.ctm_absorb_x(self.h, d, arr, chi_in, chi_out, self._adims(A), int(normalize), _ptr(nC1), _ptr(nC2), _ptr(nT)), "absorb")
        return nC1, nC2, nT

    def move(self, direction, units, chi, cfg=None, normalize=1, skip_zero_columns=False, workers=()):
        """One whole directional move in ONE native call (ctm_move): units = list of dicts, one per site, with
        "t16" (projector window, as projectors_4x4), "basis" (or None), "corners" (list of four (buffer, valid) or None),
        "absorb6" (C1, T1, T, T2, C2, A of the site's absorb) and "nb" (index of the neighbour unit along the move).
        workers: engines (own contexts and streams) on which the units of a phase run concurrently; () = serially on this engine.
        Returns per unit (P, Pt, S, nC1, nC2, nT, ncol)."""
        d = DIR_INDEX[direction] if isinstance(direction, tuple) else direction
        cfg = cfg or self.default_cfg
        arr = (MoveUnit * len(units))()
        keep, outs = [], []
        dtype = None
        for i, u in enumerate(units):
            ts, parr, ad = self._pack16(u["t16"])
            dtype = ts[0].dtype
            m = arr[i]
            for j in range(16): m.proj[j] = ts[j].data_ptr()
            for j in range(20): m.proj_adims[j] = ad[j]
            # the same argument checks as projectors_4x4 / absorb (a malformed workspace must be a NativeError, not a device write
            # out of bounds): ctm_move reads the environment with dimension chi on every leg and writes chi x chi corners
            chi_env = ts[0].shape[0]
            n = chi_env * ts[3].shape[CUT_LEG[d]] ** 2
            if chi_env != chi or u["absorb6"][0].shape[0] != chi or u["absorb6"][4].shape[0] != chi:
                raise NativeError(f"move: unit {i}: environment tensors of dimension {chi_env} / {u['absorb6'][0].shape[0]} in a move to chi = {chi} "
                                  "(ctm_move takes environments of dimension chi; use projectors_4x4 + absorb for chi-ramping)")
            b = u.get("basis")
            if b is not None:
                if not (b.is_cuda and b.device == ts[0].device and b.dtype == torch.float64 and b.is_contiguous()
                        and tuple(b.shape) == (self.warm_rows(chi, n, dtype)[0], n)):
                    raise NativeError(f"move: unit {i}: basis must come from warm_basis(chi, n, dtype)")
            m.basis = b.data_ptr() if b is not None else None
            cs = u.get("corners")
            if cs is not None:
                for buf, _ in cs:
                    if buf is not None and not (buf.is_cuda and buf.device == ts[0].device and buf.dtype == torch.float64 and buf.is_contiguous()):
                        raise NativeError(f"move: unit {i}: corner buffers must be contiguous float64 tensors on the engine's device")
            m.use_corner_cache = 1 if cs is not None else 0
            for j in range(4):
                buf, valid = cs[j] if cs is not None else (None, False)
                m.corner_buf[j] = buf.data_ptr() if buf is not None else None
                m.corner_valid[j] = int(bool(valid))
            a6 = [_chk_t(t, "absorb tensor", dtype) for t in u["absorb6"]]
            for j in range(6): m.absorb[j] = a6[j].data_ptr()
            A = a6[5]
            for j in range(5): m.absorb_adims[j] = A.shape[j]
            m.nb = int(u["nb"])
            kc = min(chi, n)
            m.n_rows = n
            D2 = A.shape[(3, 4, 1, 2)[d]] ** 2
            shapes = {UP: (kc, D2, kc), LEFT: (kc, kc, D2), DOWN: (D2, kc, kc), RIGHT: (kc, D2, kc)}
            cs_ = lambda new_axis: (kc, chi_env) if new_axis == 0 else (chi_env, kc)
            n1, n2 = self.NEW_AXIS[d]
            P, Pt, S = torch.empty((n, kc), dtype=dtype, device=self.device), torch.empty((n, kc), dtype=dtype, device=self.device), self.empty_real(kc)
            nC1, nC2 = torch.empty(cs_(n1), dtype=dtype, device=self.device), torch.empty(cs_(n2), dtype=dtype, device=self.device)
            nT = torch.empty(shapes[d], dtype=dtype, device=self.device)
            m.P, m.Pt, m.S, m.nC1, m.nC2, m.nT = P.data_ptr(), Pt.data_ptr(), S.data_ptr(), nC1.data_ptr(), nC2.data_ptr(), nT.data_ptr()
            keep.append((ts, a6))
            outs.append([P, Pt, S, nC1, nC2, nT])
        self._bind_dtype(dtype)
        wh = []
        for w in workers:
            w._bind_dtype(dtype)
            wh.append(w._handles[dtype])
        warr = (C.c_void_p * max(len(wh), 1))(*[h.value for h in wh])
        st = self.lib.ctm_move(self.h, warr, len(wh), d, len(units), arr, chi, C.byref(cfg), int(normalize), int(bool(skip_zero_columns)))
        if st == 5:
            msg = self.lib.ctm_last_error(self.h)
            raise ValueError("ctm_move: " + (msg.decode() if msg else "unsupported shapes"))
        self._ck(st, "move")
        return [tuple(o) + (int(arr[i].ncol),) for i, o in enumerate(outs)]

    # ---- C4v ------------------------------------------------------------------------------------------
    def c2x2_c4v(self, a, C_, T, open_=False):
        a, C_, T = self._bind(a, C_, T)
        chi, p, D = C_.shape[0], a.shape[0], a.shape[1]
        n = chi * D * D
        out = self.empty(n, n, p, p) if open_ else self.empty(n, n)
        self._ck(self.lib.ctm_c2x2_c4v(self.h, int(open_), _ptr(a), _ptr(C_), _ptr(T), chi, p, D, _ptr(out)), "c2x2_c4v")
        return out

    def warm_basis_c4v(self, chi, n, dtype=torch.float64):
        """Zero-filled warm-start workspace for move_c4v / truncated_eigh(..., basis=): (chi + 1 + 8) rows of length n (for complex128
        matrices the real plane followed by the imaginary plane) and one header row in which the solver keeps its adaptive state."""
        k = chi + 1 if chi < n else n
        return torch.zeros((2 if dtype.is_complex else 1) * min(n, k + 8) + 1, n, dtype=torch.float64, device=self.device)   # + header row

    def move_c4v(self, a, C_, T, cfg=None, basis=None, normalize=1):
        a, C_, T = self._bind(a, C_, T)
        chi, p, D = C_.shape[0], a.shape[0], a.shape[1]
        nC, nT, Dv = self.empty(chi, chi), self.empty(chi, chi, D * D), self.empty_real(chi)
        cfg = cfg or self.cfg(eps_multiplet=1e-12)
        if basis is not None:
            n = chi * D * D
            k = chi + 1 if chi < n else n
            if not (basis.is_cuda and basis.dtype == torch.float64 and basis.is_contiguous()
                    and tuple(basis.shape) == ((2 if a.dtype.is_complex else 1) * min(n, k + 8) + 1, n)):
                raise NativeError("move_c4v: basis must come from warm_basis_c4v(chi, n, dtype)")
        self._ck(self.lib.ctm_move_c4v_x(self.h, _ptr(a), _ptr(C_), _ptr(T), chi, p, D, C.byref(cfg), int(normalize), _ptr(nC), _ptr(nT), _ptr(Dv),
                                         _ptr(basis) if basis is not None else None), "move_c4v")
        return nC, nT, Dv

    def rdm_c4v(self, which, a, C_, T):
        a, C_, T = self._bind(a, C_, T)
        chi, p, D = C_.shape[0], a.shape[0], a.shape[1]
        out = self.empty(*([p] * (8 if which == 3 else 4)))
        self._ck(self.lib.ctm_rdm_c4v(self.h, which, _ptr(a), _ptr(C_), _ptr(T), chi, p, D, _ptr(out)), "rdm_c4v")
        return out

    # ---- RDMs ------------------------------------------------------------------------------------------
    def rdm2x2(self, tensors16):
        ts, arr, ad = self._pack16(tensors16)
        p = ts[3].shape[0]
        out = self.empty(*([p] * 8))
        self._ck(self.lib.ctm_rdm2x2(self.h, arr, ts[0].shape[0], ad, _ptr(out)), "rdm2x2")
        return out

    def rdm2x2_part(self, tensors16, lo0, lo1):
        """Raw block R[(s0 t0 s1 t1), lo0:lo1] of the plaquette contraction (lower-half slices lo0..lo1-1 of p^4)."""
        ts, arr, ad = self._pack16(tensors16)
        p = ts[3].shape[0]
        out = self.empty(p ** 4, lo1 - lo0)
        self._ck(self.lib.ctm_rdm2x2_part(self.h, arr, ts[0].shape[0], ad, int(lo0), int(lo1), _ptr(out)), "rdm2x2_part")
        return out

    @staticmethod
    def rdm2x2_from_parts(R, p):
        """R[s0 t0 s1 t1 ; s2 t2 s3 t3] (p^4 x p^4, the parts side by side; s2 = site coord+y, s3 = coord+x+y)
        -> rdm[s0 s1 s2 s3 ; t0 t1 t2 t3] (rdm.py:1581-1588)."""
        return R.reshape([p] * 8).permute(0, 2, 4, 6, 1, 3, 5, 7).contiguous()

    def rdm1x1(self, tensors9):
        ts = self._bind(*tensors9)
        arr = (C.c_void_p * 9)(*[t.data_ptr() for t in ts])
        a = ts[8]
        out = self.empty(a.shape[0], a.shape[0])
        self._ck(self.lib.ctm_rdm1x1(self.h, arr, ts[0].shape[0], self._adims(a), _ptr(out)), "rdm1x1")
        return out

    def _rdm2(self, fn, tensors12, name):
        ts = self._bind(*tensors12)
        arr = (C.c_void_p * 12)(*[t.data_ptr() for t in ts])
        a0, a1 = ts[5], ts[11]
        ad = (C.c_int * 10)(*(list(a0.shape) + list(a1.shape)))
        p = a0.shape[0]
        out = self.empty(p, p, p, p)
        self._ck(fn(self.h, arr, ts[0].shape[0], ad, _ptr(out)), name)
        return out

    def rdm2x1(self, tensors12):
        return self._rdm2(self.lib.ctm_rdm2x1, tensors12, "rdm2x1")

    def rdm1x2(self, tensors12):
        return self._rdm2(self.lib.ctm_rdm1x2, tensors12, "rdm1x2")

    def init_piece(self, kind, a):
        a = self._bind(a)
        kept = [(3, 4), (2, 3), (1, 2), (1, 4), (2, 3, 4), (1, 3, 4), (1, 2, 4), (1, 2, 3)][kind]
        out = self.empty(*[a.shape[k] ** 2 for k in kept])
        self._ck(self.lib.ctm_init_piece(self.h, kind, _ptr(a), self._adims(a), _ptr(out)), "init_piece")
        return out


_engines = {}


def _shutdown():
    """Release every native context while the HIP runtime is still alive (interpreter exit order is otherwise arbitrary)."""
    try:
        import units
        units.shutdown()
    except Exception:
        pass
    for e in list(_engines.values()):
        try:
            e.close()
        except Exception:
            pass
    _engines.clear()


import atexit
atexit.register(_shutdown)


def engine(device=None):
    """Process-wide engine per device (created lazily)."""
    if not torch.cuda.is_available():
        raise NativeError("no HIP device visible: the CTM engine runs only on the GPU (no CPU fallback)")
    if device is None:
        idx = torch.cuda.current_device()
    else:
        d = torch.device(device)
        if d.type != "cuda":
            raise NativeError(f"the CTM engine runs only on HIP devices, not on '{device}'")
        idx = d.index if d.index is not None else torch.cuda.current_device()
    if idx not in _engines:
        _engines[idx] = Engine(torch.device("cuda", idx))
        # development hook: engine options from the environment, "key=value,key=value" (same keys as Engine.set_option)
        for kv in filter(None, os.environ.get("CTM_ENGINE_OPTS", "").split(",")):
            k, v = kv.split("=")
            _engines[idx].set_option(k.strip(), float(v))
    return _engines[idx]
