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
    """ONE pool per device.  Slots (stream + native worker context with its workspace arena) are created on demand and never
    duplicated: a call that wants fewer units in flight than the pool has slots is limited to the LOWEST slots (`limit`), so the
    number of arenas that ever grow is the maximum any call asked for -- not the sum over the different widths a run used (a full-rank
    run at n >= 8192 alternates between 2 and 4 units in flight; two pools meant six contexts of ~30 GB high-water arena each)."""
    MAX_SLOTS = 16

    def __init__(self, main_engine):
        self.main = main_engine
        self.device = main_engine.device
        self.streams = []
        self.engines = []
        self.pool = ThreadPoolExecutor(max_workers=self.MAX_SLOTS, thread_name_prefix="ctm-unit")      # (threads start lazily)
        self.busy = set()
        self.cv = threading.Condition()

    def _take(self, limit):
        with self.cv:
            while True:
                for slot in range(limit):
                    if slot not in self.busy:
                        self.busy.add(slot)
                        while len(self.streams) <= slot:
                            self.streams.append(torch.cuda.Stream(self.device)); self.engines.append(None)
                        return slot
                self.cv.wait()

    def ensure(self, n):
        """Streams and worker engines of the first n slots (for callers that hand the contexts to the library: Engine.move)."""
        with self.cv:
            while len(self.streams) < n:
                self.streams.append(torch.cuda.Stream(self.device)); self.engines.append(None)
            for slot in range(n):
                if self.engines[slot] is None:
                    with torch.cuda.device(self.device), torch.cuda.stream(self.streams[slot]):
                        self.engines[slot] = self.main.spawn_worker()
                        backend.register_stream_engine(self.streams[slot], self.engines[slot])
            return self.engines[:n]

    def _give(self, slot):
        with self.cv:
            self.busy.discard(slot)
            self.cv.notify_all()

    def _run(self, fn, item, ev0, limit, delay=0.0):
        slot = self._take(limit)                                     # an idle worker context among the first `limit`
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
            self._give(slot)

    def map(self, fn, items, limit, stagger=0.0):
        """[fn(item) for item in items], at most `limit` at a time, each on its own stream/context; a worker takes the next
        item as soon as it has issued its previous one (no barrier between groups of `limit` items)."""
        items = list(items)
        limit = max(1, min(int(limit), self.MAX_SLOTS))
        main_stream = torch.cuda.current_stream(self.device)
        ev0 = torch.cuda.Event()
        ev0.record(main_stream)
        # `stagger` seconds between the starts of the first `limit` items: units whose latency-bound phases (orthogonalisations,
        # Ritz extraction) would otherwise coincide run out of phase, so that one unit's small kernels overlap another's corner passes
        futs = [self.pool.submit(self._run, fn, it, ev0, limit, stagger * i if i < limit else 0.0) for i, it in enumerate(items)]
        outs = []
        for f in futs:
            out, ev = f.result()
            main_stream.wait_event(ev)
            outs.append(out)
        return outs


class PoolView:
    """What pool_for hands out: the device's pool with the number of units this caller may have in flight."""

    def __init__(self, pool, n):
        self.pool, self.n = pool, n

    def map(self, fn, items, stagger=0.0):
        return self.pool.map(fn, items, self.n, stagger=stagger)

    def engines(self):
        return self.pool.ensure(self.n)


_pools = {}
_lock = threading.Lock()


def _held_bytes(pool, nslots):
    """Workspace the first `nslots` worker contexts of the device's pool hold right now (arenas are high-water: they do not shrink
    between calls)."""
    tot = 0.0
    for e in (pool.engines[:nslots] if pool is not None else []):
        if e is not None:
            tot += e.own_stat("arena_total")
    return tot


def pool_for(engine, nunits, n, is_complex, est_bytes=None, large_n_units=None):
    """The device's UnitPool limited to the number of concurrent units of fused dimension n that fit, or None when concurrency is
    off / pointless (stand-in engines, a single unit, or not enough HBM for one workspace arena per unit).  est_bytes overrides the
    per-unit workspace estimate of a sweep unit.  The HBM budget (half the device) is counted across everything the pool's
    contexts already hold, not per call."""
    if nunits < 2 or not hasattr(engine, "spawn_worker"):
        return None
    total = torch.cuda.get_device_properties(engine.device).total_memory
    est = est_bytes if est_bytes is not None else 14.0 * n * n * 8 * (2 if is_complex else 1)   # corners, products, work
    # budget: half the device, and not more than what is free right now plus what the pool's contexts already hold (cached enlarged
    # corners and the environment are not the pool's to use)
    with _lock:
        held = _held_bytes(_pools.get(engine.device.index), UnitPool.MAX_SLOTS)
    budget = min(0.5 * total, 0.9 * (torch.cuda.mem_get_info(engine.device)[0] + held))
    nw = int(min(nunits, max(1, budget // max(est, 1.0))))
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
    key = engine.device.index
    with _lock:
        pool = _pools.get(key)
        if pool is None:
            pool = _pools[key] = UnitPool(engine)
        # contexts beyond the slots this call uses keep their arenas: when they and this call's estimate do not fit the budget
        # together, give the idle ones back (between calls every arena stack is empty)
        idle = pool.engines[nw:]
        if idle and _held_bytes(pool, len(pool.engines)) - _held_bytes(pool, nw) + nw * est > budget:
            with pool.cv:
                if not pool.busy:
                    for e in idle:
                        if e is not None:
                            e.trim()
        return PoolView(pool, nw)


def shutdown():
    """Stop the worker threads (their engines are owned and closed by the main engine)."""
    with _lock:
        for pool in _pools.values():
            pool.pool.shutdown(wait=True)
        _pools.clear()
