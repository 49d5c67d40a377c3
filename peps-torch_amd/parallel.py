"""Multi-GPU sharding of the per-site units of a directional move and of per-site RDMs.

One process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI on ROCm, "gloo" in the
CPU tests).  Inside one ctm_MOVE all sites' projector computations read the SAME old environment and
all absorbs read the old environment plus the full projector set (reference ctm/generic/ctmrg.py:
238-275), so a move is two embarrassingly parallel phases over sites with one exchange after each:
  phase A  rank r: (P, Pt) of its sites           -> all-gather of P, Pt   (2 n chi doubles / site)
  phase B  rank r: (nC1, nC2, nT) of its sites    -> all-gather            ((2 chi^2 + chi^2 D^2) / site)
The four directions stay sequential (move k+1 reads what move k wrote).  Environment tensors are
replicated on every rank, so after the second exchange all ranks hold identical environments.
"""
import torch

try:
    import torch.distributed as dist
except Exception:                                      # pragma: no cover
    dist = None


# Test-only switch (tests/test_gpu_dist.py, tools/check_rccl_world1.py): treat an initialised process group of ONE rank as distributed,
# so that every collective of this module (RCCL init, device buffers, the (re,im) view of complex128, the in-place all-gather,
# sub-groups) executes on the single GPU the build loop has.  Never set by the product path.
single_rank_is_distributed = False


def is_distributed():
    return dist is not None and dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or single_rank_is_distributed)


def world():
    if is_distributed():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def owner_of(index, nranks):
    """Static round-robin ownership of unit `index` (site number inside the unit cell)."""
    return index % nranks


def unit_group_size(nitems):
    """Ranks per unit when there are MORE ranks than units and they divide evenly (8 GPUs, 4-site cell: SURVEY 8e "pair (2r, 2r+1) splits unit
    r" -- here the pair of unit i is {i, i + nitems}, the same grouping as site_groups): those ranks SHARE the unit (Engine.set_group: every
    corner pass split by output columns inside the native solver).  1 otherwise.  Groups of two only (include/ctm_hip.h)."""
    _, n = world()
    return 2 if (is_distributed() and nitems > 0 and n == 2 * nitems) else 1


def unit_group(index, nitems):
    _, n = world()
    return [r for r in range(n) if r % nitems == index]


def my_units(items, shared=False):
    """Subset of `items` (ordered) owned by this rank; shared (unit_group_size(len(items)) == 2): the unit this rank's group works on."""
    rank, n = world()
    if shared:
        return [x for i, x in enumerate(items) if rank % len(items) == i]
    return [x for i, x in enumerate(items) if owner_of(i, n) == rank]


def site_groups(nitems):
    """Ranks that work on each of `nitems` independent items (per-site RDMs).  Up to one rank per item: round-robin owners
    (one-member groups).  MORE ranks than items (8 GPUs, 4-site cell: SURVEY 8e): item i gets the ranks {r : r mod nitems == i},
    which share its contraction (rdm._rdm2x2_raw splits the p^4 lower-half slices among them)."""
    _, n = world()
    if n <= nitems:
        return [[owner_of(i, n)] for i in range(nitems)]
    return [[r for r in range(n) if r % nitems == i] for i in range(nitems)]


_group_cache = {}


def _process_group(members):
    """torch.distributed group of `members` (created collectively: every rank must ask for the same groups in the same order)."""
    key = tuple(members)
    if key not in _group_cache:
        _group_cache[key] = dist.new_group(ranks=list(members))
    return _group_cache[key]


def prepare_groups(groups):
    """Collective creation of the process groups of `groups` (all ranks call this with the same list)."""
    if is_distributed():
        for g in groups:
            if len(g) > 1 or single_rank_is_distributed:
                _process_group(g)


def allreduce_sum_group(t, members):
    """Sum of `t` over the ranks in `members` (complex128 travels as its (re,im) float64 view)."""
    if not is_distributed() or (len(members) < 2 and not single_rank_is_distributed):
        return t
    buf = torch.view_as_real(t.contiguous()) if t.is_complex() else t.contiguous()
    dist.all_reduce(buf, group=_process_group(members))
    return torch.view_as_complex(buf) if t.is_complex() else buf


def owners_shifted(coords, vertex_to_site, shift):
    """Owner rank of every unit when unit `c` is given to the round-robin owner of the site at c + shift.  Used by the
    projector phase: with the per-direction shifts of ctmrg._OWNER_SHIFT the two enlarged corners a window can reuse from the
    previous move (corner cache) were computed on the SAME rank."""
    _, n = world()
    idx = {c: i for i, c in enumerate(coords)}
    return [owner_of(idx[vertex_to_site((c[0] + shift[0], c[1] + shift[1]))], n) for c in coords]


# ---- time spent in the exchanges (bench.py: phase_s["comm"]) ----------------------------------------------------------------
# Event pairs on the CURRENT stream around every collective of exchange(): torch makes the current stream wait for the collective,
# so the second event fires when the data has arrived.  Read (and the events released) by comm_time_s(); no host synchronisation here.
_comm_events = []
_comm_host_s = 0.0
comm_timing = False


def _comm_begin(like):
    if not comm_timing:
        return None
    if like.is_cuda:
        e = torch.cuda.Event(enable_timing=True); e.record(torch.cuda.current_stream(like.device)); return e
    import time
    return time.perf_counter()


def _comm_end(like, e0):
    global _comm_host_s
    if e0 is None:
        return
    if like.is_cuda:
        e1 = torch.cuda.Event(enable_timing=True); e1.record(torch.cuda.current_stream(like.device)); _comm_events.append((e0, e1))
    else:
        import time
        _comm_host_s += time.perf_counter() - e0


def comm_time_s(reset=True):
    """Seconds this rank spent in exchange() collectives since the last reset (device time between the bracketing events)."""
    global _comm_host_s
    if _comm_events:
        _comm_events[-1][1].synchronize()
    tot = _comm_host_s + 1e-3 * sum(a.elapsed_time(b) for a, b in _comm_events)
    if reset:
        _comm_events.clear(); _comm_host_s = 0.0
    return tot


def exchange(local, keys, shapes, like, owners=None):
    """All ranks end up with `{key: tensor}` for every key of `keys` (ordered list, identical on all
    ranks); `local` holds the entries this rank computed; `shapes[key]` is known to every rank (it
    follows from chi and the bond dimensions), `like` supplies dtype/device, `owners[i]` (default i mod ranks) is the rank that
    computed keys[i].  When all tensors of the
    exchange have one shape and every rank owns the same number of them (uniform-D cell, #sites a
    multiple of #ranks) they travel as ONE all-gather into a preallocated (ranks x per-rank x shape) buffer (few, large messages:
    xGMI is point-to-point, per-link bound) whose slices ARE the returned tensors -- no stacked send copy, no per-rank receive
    list: with one tensor per rank the tensor itself is the send buffer, with several they are written once into this rank's
    slice of the receive buffer and gathered in place.  Otherwise per-key broadcasts from the owner."""
    if not is_distributed():
        return dict(local)
    rank, n = world()
    out = dict(local)
    own = owners if owners is not None else [owner_of(i, n) for i in range(len(keys))]
    owned = [[k for i, k in enumerate(keys) if own[i] == r] for r in range(n)]
    uniform = len({len(o) for o in owned}) == 1 and len({tuple(shapes[k]) for k in keys}) == 1 and len(owned[0]) > 0
    cx = like.dtype.is_complex       # RCCL has no complex type: complex128 travels as its (re,im) float64 view
    if uniform:
        per, shp = len(owned[0]), tuple(shapes[keys[0]])
        recv = torch.empty((n, per) + shp, dtype=like.dtype, device=like.device)
        rbuf = torch.view_as_real(recv) if cx else recv
        if per == 1:
            t = local[owned[rank][0]].contiguous()
            send = (torch.view_as_real(t) if cx else t).reshape(rbuf.shape[1:])
        else:
            for j, k in enumerate(owned[rank]):
                recv[rank, j].copy_(local[k])
            send = rbuf[rank]                                   # in-place all-gather: my slice of the receive buffer
        e0 = _comm_begin(like)
        # output = concatenation of the ranks' inputs along dim 0 (the form both RCCL and gloo accept)
        dist.all_gather_into_tensor(rbuf.view((n * per,) + tuple(rbuf.shape[2:])), send)
        _comm_end(like, e0)
        for r in range(n):
            for j, k in enumerate(owned[r]):
                out[k] = recv[r, j]
        return out
    e0 = _comm_begin(like)
    for i, k in enumerate(keys):
        src = own[i]
        t = local[k].contiguous() if src == rank else torch.empty(tuple(shapes[k]), dtype=like.dtype, device=like.device)
        dist.broadcast(torch.view_as_real(t) if cx else t, src)
        out[k] = t
    _comm_end(like, e0)
    return out


def allreduce_min_int_group(x, members, device):
    """Minimum of an integer over the ranks in `members` (a process group created by prepare_groups)."""
    if not is_distributed() or (len(members) < 2 and not single_rank_is_distributed):
        return int(x)
    t = torch.tensor([int(x)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=_process_group(members))
    return int(t.item())


def allreduce_sum_scalar(x, device):
    if not is_distributed():
        return x
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t)
    return float(t.item())


def allreduce_max_int(x, device):
    if not is_distributed():
        return int(x)
    t = torch.tensor([int(x)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


class _ExchangeAD(torch.autograd.Function):
    """exchange() as an autograd node.  Every rank runs the same program on replicated tensors, so the "graph" of a distributed
    differentiable run is the union of the ranks' local graphs joined at the exchanges: the output `key` on rank r is a copy of the
    tensor its owner computed, hence the cotangent of that tensor is the SUM over ranks of the cotangents of their copies -- one
    all-reduce of the stacked cotangents, of which a rank keeps the slices it owns.  (With every rank seeding its own replica of the
    loss this differentiates sum_r L_r = R L: average_grads() divides by R.)"""

    @staticmethod
    def forward(ctx, keys, shapes, like, owners, *local_tensors):
        rank, n = world()
        own = owners if owners is not None else [owner_of(i, n) for i in range(len(keys))]
        mine = [k for i, k in enumerate(keys) if own[i] == rank]
        out = exchange({k: t.detach() for k, t in zip(mine, local_tensors)}, keys, shapes, like, owners=owners)
        ctx.mine_idx = [i for i, k in enumerate(keys) if own[i] == rank]
        return tuple(out[k].clone() if own[i] == rank else out[k] for i, k in enumerate(keys))

    @staticmethod
    def backward(ctx, *gouts):
        flat = torch.cat([(torch.view_as_real(g.contiguous()) if g.is_complex() else g.contiguous()).reshape(-1) for g in gouts])
        dist.all_reduce(flat)
        grads, off = [], 0
        for i, g in enumerate(gouts):
            m = g.numel() * (2 if g.is_complex() else 1)
            if i in ctx.mine_idx:
                piece = flat[off:off + m]
                grads.append(torch.view_as_complex(piece.reshape(g.shape + (2,))) if g.is_complex() else piece.reshape(g.shape))
            off += m
        return (None, None, None, None) + tuple(grads)


def exchange_ad(local, keys, shapes, like, owners=None):
    """Differentiable exchange(): same result, and gradients flow back to the rank that computed each tensor."""
    if not is_distributed():
        return dict(local)
    rank, n = world()
    own = owners if owners is not None else [owner_of(i, n) for i in range(len(keys))]
    mine = [k for i, k in enumerate(keys) if own[i] == rank]
    outs = _ExchangeAD.apply(list(keys), shapes, like, owners, *[local[k] for k in mine])
    return dict(zip(keys, outs))


def average_grads(tensors):
    """After loss.backward() of a distributed differentiable run: the gradient of the (replicated) loss with respect to the replicated
    parameters is the mean over ranks of the local .grad fields (see _ExchangeAD)."""
    if not is_distributed():
        return
    _, n = world()
    for t in tensors:
        if t.grad is None:
            t.grad = torch.zeros_like(t)
        buf = torch.view_as_real(t.grad) if t.grad.is_complex() else t.grad
        dist.all_reduce(buf)
        t.grad.div_(n)
