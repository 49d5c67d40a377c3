"""Concurrent execution of the independent (site, direction) units of one directional move on ONE GPU.

Inside a move the per-site projector constructions (and, after them, the per-site absorptions) only read the old
environment (reference ctm/generic/ctmrg.py:238-275), so they are independent.  At small and medium n = chi D^2 a
single unit cannot fill 256 CUs (its chi-truncation has latency-bound stages: one-workgroup LDS eigensolves, host
decisions between half steps), so the units of a phase are issued from separate host threads, each with its own
native context, workspace arena and HIP stream; the GPU overlaps them.  Stream order: every worker stream waits for
an event recorded on the caller's stream before it starts, and the caller's stream waits for every worker's final
event, so each phase is bracketed by full cross-stream barriers (which also makes cross-stream reuse of torch's
cached blocks safe: tensors are only released at phase boundaries)."""
import threading
import time
import queue
from concurrent.futures import ThreadPoolExecutor
import os
import torch
import backend


class UnitPool:
    def __init__(self, main_engine, nworkers):
        self.main = main_engine
        self.device = main_engine.device
        self.n = nworkers
        self.streams = [torch.cuda.Stream(self.device) for _ in range(nworkers)]
        self.engines = [None] * nworkers
        self.pool = ThreadPoolExecutor(max_workers=nworkers, thread_name_prefix="ctm-unit")
        self.free = queue.Queue()
        for slot in range(nworkers):
            self.free.put(slot)

    def _run(self, fn, item, ev0, delay=0.0):
        slot = self.free.get()                                       # a worker context that is idle right now
        try:
            if delay > 0.0:
                time.sleep(delay)
            s = self.streams[slot]
            with torch.cuda.device(self.device), torch.cuda.stream(s):
                s.wait_event(ev0)
                if self.engines[slot] is None:
                    self.engines[slot] = self.main.spawn_worker()    # binds the current (= worker) stream
                    backend.register_stream_engine(s, self.engines[slot])
                backend.set_thread_engine(self.engines[slot])
                try:
                    out = fn(item)
                finally:
                    backend.set_thread_engine(None)
                ev = torch.cuda.Event()
                ev.record(s)
            return out, ev
        finally:
            self.free.put(slot)

    def map(self, fn, items, stagger=0.0):
        """[fn(item) for item in items], at most `nworkers` at a time, each on its own stream/context; a worker takes the next
        item as soon as it has issued its previous one (no barrier between groups of `nworkers` items)."""
        items = list(items)
        main_stream = torch.cuda.current_stream(self.device)
        ev0 = torch.cuda.Event()
        ev0.record(main_stream)
        # `stagger` seconds between the starts of the first `nworkers` items: units whose latency-bound phases (orthogonalisations,
        # Ritz extraction) would otherwise coincide run out of phase, so that one unit's small kernels overlap another's corner passes
        futs = [self.pool.submit(self._run, fn, it, ev0, stagger * i if i < self.n else 0.0) for i, it in enumerate(items)]
        outs = []
        for f in futs:
            out, ev = f.result()
            main_stream.wait_event(ev)
            outs.append(out)
        return outs


_pools = {}
_lock = threading.Lock()


def pool_for(engine, nunits, n, is_complex, est_bytes=None, large_n_units=None):
    """UnitPool sized for `nunits` concurrent units of fused dimension n, or None when concurrency is off / pointless
    (stand-in engines, a single unit, or not enough HBM for one workspace arena per unit).  est_bytes overrides the
    per-unit workspace estimate of a sweep unit."""
    if nunits < 2 or not hasattr(engine, "spawn_worker"):
        return None
    total = torch.cuda.get_device_properties(engine.device).total_memory
    est = est_bytes if est_bytes is not None else 14.0 * n * n * 8 * (2 if is_complex else 1)   # corners, products, work
    nw = int(min(nunits, max(1, (0.5 * total) // max(est, 1.0))))
    if n >= 8192:
        # kernels of this size fill the chip on their own, but the latency-bound stages of a unit (block orthogonalisations, the
        # dense SVD of the Ritz matrix: ~40 % of a full-rank unit's time, a few workgroups wide) only overlap with OTHER units'
        # latency-bound stages -- the units of a move run in lockstep.  All four units of a move in flight: those stages are
        # paid once per move instead of twice (full-rank D = 8 chi = 256: 3.71 -> 3.26 s/sweep).  Needs one hardware queue per
        # stream (GPU_MAX_HW_QUEUES, see backend.py): with the default of four queues the four worker streams share two of them
        # and run pairwise serialised (measured: no gain at all).
        nw = min(nw, int(os.environ.get("CTM_LARGE_N_UNITS", 4)) if large_n_units is None else large_n_units)
    nw = min(nw, int(os.environ.get("CTM_MAX_CONCURRENT_UNITS", nw)))       # experiment knob
    if nw < 2:
        return None
    key = (engine.device.index, nw)
    with _lock:
        if key not in _pools:
            _pools[key] = UnitPool(engine, nw)
        return _pools[key]


def shutdown():
    """Stop the worker threads (their engines are owned and closed by the main engine)."""
    with _lock:
        for pool in _pools.values():
            pool.pool.shutdown(wait=True)
        _pools.clear()
