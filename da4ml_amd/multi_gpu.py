"""One process per GPU: sharding of independent CMVM solves over the ranks of a ``torch.distributed`` job.

The path shards at the level of independent units (matrices, decompose_dc candidates, model layers): every rank solves
its contiguous shard with no data-path collective.  Collectives are used only for the exchange
steps that really exist: the max-over-ranks timing of the benchmark, the arg-min over candidate costs, and gathering
results to rank 0.  Backend ``nccl`` (= RCCL over xGMI on ROCm) when a GPU is present, ``gloo`` on CPU-only hosts
(used by the CPU tests with world_size 2).
"""

from __future__ import annotations

import os


def env_rank() -> tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment; (0, 1, 0) when launched directly."""
    return int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))


_rccl_ids: dict = {}  # (world, rank, device) -> the RCCL unique id of the process group (solve_column_sharded(transport='rccl'))
_unpinned_cores: list[int] | None = None  # affinity mask of the process before pin_rank_to_core_slice first narrowed it
_pinned_as: tuple | None = None          # (local rank, local world, cores) of the pinning in force


def pin_rank_to_core_slice(local: int, local_world: int) -> list[int] | None:
    """One process per GPU on one host: give local rank ``local`` of ``local_world`` its contiguous slice of the cores this
    process may run on.  The library's host pool (adder trees, staging; ``csrc/cmvm_host.cc``) sizes itself from the affinity
    mask, and the thread that queues the greedy loop's launches -- latency-critical: two launches per group and step -- is not
    preempted by seven other ranks' pools: without this, eight ranks start eight pools of up to 256 threads on the same cores.
    Affects the calling thread and every thread it starts afterwards; ``DA4ML_PIN_RANKS=0`` turns it off."""
    if local_world <= 1 or os.environ.get('DA4ML_PIN_RANKS', '1') == '0' or not hasattr(os, 'sched_setaffinity'):
        return None
    # Idempotent: init() runs from every sharded solve.  The slice is always cut from the mask the process had BEFORE its first
    # pinning (slicing the already narrowed mask again would leave 1 / local_world of it per call: 32, 4, 1 cores ...), and a
    # repeated call with the same arguments is a no-op.
    global _unpinned_cores, _pinned_as
    if _unpinned_cores is None:
        _unpinned_cores = sorted(os.sched_getaffinity(0))
    if _pinned_as is not None and _pinned_as[:2] == (local, local_world):
        return _pinned_as[2]
    cores = _unpinned_cores
    lo, hi = len(cores) * local // local_world, len(cores) * (local + 1) // local_world
    mine = cores[lo:hi] or cores[local % len(cores) : local % len(cores) + 1]
    os.sched_setaffinity(0, mine)
    _pinned_as = (local, local_world, mine)
    return mine


def init(backend: str | None = None):
    """Initialise ``torch.distributed`` from the environment; returns (rank, world, local_rank, device)."""
    rank, world, local = env_rank()
    pin_rank_to_core_slice(local, int(os.environ.get('LOCAL_WORLD_SIZE', world)))  # before torch and the library start their threads
    import torch
    import torch.distributed as dist

    use_gpu = torch.cuda.is_available()
    device = torch.device(f'cuda:{local}') if use_gpu else torch.device('cpu')
    if use_gpu:
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:  # the ranks cannot agree on a free port by themselves: the launcher picks it
            raise RuntimeError('WORLD_SIZE > 1 but MASTER_PORT is not set: start the ranks with torch.distributed.run or `python bench.py --gpus N`')
        dist.init_process_group(backend or ('nccl' if use_gpu else 'gloo'), rank=rank, world_size=world)
    return rank, world, local, device


def shutdown():
    """Tear the process group down (after a last barrier) so that no backend thread outlives the interpreter:
    a rank that exits with a live group can abort in the backend's destructors ('terminate called without an active
    exception'), which a launcher reports as a failed job."""
    import torch.distributed as dist

    try:
        if _rccl_ids:  # communicators of the library's own RCCL transport: destroyed while the group is alive (every rank gets here)
            from . import _binary

            _rccl_ids.clear()
            _binary.rccl_shutdown()
    finally:  # (a failing RCCL teardown must not leave the process group alive: that exit is what this function exists to prevent)
        if dist.is_available() and dist.is_initialized():
            try:
                dist.barrier()
            finally:
                dist.destroy_process_group()


def shard_bounds(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous balanced shard [lo, hi) of ``n_items`` independent units for ``rank``."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def barrier():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    """All-reduce(max) of a scalar (the benchmark's elapsed time)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def argmin_first(costs_local: list[float], lo: int, n_total: int, device=None) -> int:
    """Global index of the first strict minimum of per-candidate costs that are sharded over ranks
    (the reference's candidate selection rule, api.cc:243-247).  ``costs_local`` are this rank's entries [lo, lo+len)."""
    import torch
    import torch.distributed as dist

    full = torch.full((n_total,), float('inf'), dtype=torch.float32, device=device or 'cpu')
    if costs_local:
        full[lo : lo + len(costs_local)] = torch.tensor(costs_local, dtype=torch.float32, device=full.device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(full, op=dist.ReduceOp.MIN)
    best = 0
    vals = full.tolist()
    for i in range(1, n_total):
        if vals[i] < vals[best]:
            best = i
    return best


def gather_to_rank0(obj):
    """Gather picklable per-rank results on rank 0 (list ordered by rank); other ranks get None."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return [obj]
    out = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(obj, out, dst=0)
    return out


def estimate_chain_cost(kernel) -> float:
    """Cheap host-side proxy for the greedy-loop work of one matrix: the number of initial digit pairs
    ``sum_col C(d_col, 2)`` with ``d_col`` the non-zero CSD digits of a column (P_init; the number of
    non-zero digits of the signed-digit recoding of ``n`` is ``popcount((3n ^ n) >> 1)``).  Only used to balance shards:
    it never influences a result."""
    import numpy as np

    k = np.asarray(kernel, dtype=np.float64)
    if k.size == 0:
        return 0.0
    nz = np.abs(k[k != 0])
    if nz.size == 0:
        return 1.0
    # scale to integers by the finest power of two present (bounded: this is an estimate, not arithmetic)
    frac = 0
    while frac < 24 and np.any(nz * 2.0**frac != np.floor(nz * 2.0**frac)):
        frac += 1
    n = np.minimum(np.abs(k) * 2.0**frac, 2.0**40).astype(np.int64)
    x = ((3 * n) ^ n) >> 1
    digits = np.zeros(k.shape, dtype=np.int64)
    while np.any(x):
        digits += x & 1
        x >>= 1
    d_col = digits.sum(axis=0).astype(np.float64)
    return float(np.sum(d_col * (d_col - 1.0) / 2.0)) + float(k.shape[0] + k.shape[1])


def balanced_shards(costs, world: int) -> list[list[int]]:
    """Deterministic longest-processing-time assignment of units to ranks: units by decreasing cost (ties: lower index
    first), each to the currently least loaded rank (ties: lower rank).  Every rank computes the same table from the same
    costs, so no exchange is needed; within a rank the units keep their original order."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * world
    shards: list[list[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda j: (load[j], j))
        shards[r].append(i)
        load[r] += float(costs[i])
    return [sorted(s) for s in shards]


def solve_many_sharded(kernels, balance: str = 'cost', solver_many=None, **opts):
    """Solve independent matrices over all ranks: rank r solves its shard on its own GPU; rank 0 returns all results in
    the order of ``kernels`` (other ranks: None).  No data-path collective, one result gather.

    ``balance='cost'`` (default) assigns matrices by estimated chain work (``estimate_chain_cost`` + ``balanced_shards``):
    the layers of a model differ in size by orders of magnitude (BASELINE config C5), and a contiguous split by count
    leaves most GPUs idle behind the one that drew the big layer.  ``balance='count'`` is the contiguous split (what
    ``bench.py`` uses for its identical matrices).  ``solver_many`` defaults to the HIP path; the CPU tests inject a
    stand-in."""
    rank, world, local, _ = init()
    if solver_many is None:
        from . import _binary

        if _binary.device_count() > 0:
            _binary.set_device(local % _binary.device_count())
        solver_many = _binary.solve_many
    if balance == 'cost':
        shards = balanced_shards([estimate_chain_cost(k) for k in kernels], world)
    elif balance == 'count':
        shards = [list(range(*shard_bounds(len(kernels), r, world))) for r in range(world)]
    else:
        raise ValueError(f"balance must be 'cost' or 'count', not {balance!r}")
    mine = shards[rank]
    per_kernel = {k: v for k, v in opts.items() if k in ('qintervals', 'latencies') and v is not None}
    local_opts = {k: v for k, v in opts.items() if k not in per_kernel}
    for k, v in per_kernel.items():  # one entry per matrix: follow the shard
        local_opts[k] = [v[i] for i in mine]
    solved = solver_many([kernels[i] for i in mine], **local_opts) if mine else []
    parts = gather_to_rank0(list(zip(mine, solved)))
    if parts is None:
        return None
    out = [None] * len(kernels)
    for part in parts:
        for i, p in part:
            out[i] = p
    return out


def candidate_list(n_in: int, hard_dc: int = -1) -> tuple[list[int], int]:
    """The ``decompose_dc`` candidates of a searching solve and the ``hard_dc`` each one is run with
    (reference api.cc:190-201: ``_hard_dc = hard_dc if hard_dc >= 0 else 1e9``; candidates -1 .. min(_hard_dc,
    ceil(log2f(n_in))))."""
    import numpy as np

    eff = hard_dc if hard_dc >= 0 else 1_000_000_000
    top = min(eff, int(np.ceil(np.log2(np.float32(n_in)))))
    return list(range(-1, top + 1)), eff


def pipeline_cost_f32(pipe) -> float:
    """Candidate cost exactly as the reference accumulates it: float32, sequentially in op order over both stages
    (api.cc:222-229)."""
    import numpy as np

    acc = np.float32(0.0)
    for sol in pipe.solutions:
        costs = sol.ops._column('cost') if hasattr(sol.ops, '_column') else [op.cost for op in sol.ops]  # (a lazy OpList: no Op is built)
        for c in costs:
            acc = np.float32(acc + np.float32(c))
    return float(acc)


def solve_candidates_sharded(kernel, method0: str = 'wmc', method1: str = 'auto', hard_dc: int = -1, qintervals=None, latencies=None,
                             adder_size: int = -1, carry_size: int = -1, solver=None):
    """One searching ``solve`` (``search_all_decompose_dc=True``) with its candidates sharded over the ranks
    (BASELINE config C4, candidate-sharded variant).

    Candidate ``i`` of the reference's search is exactly ``solve(..., hard_dc=_hard_dc, decompose_dc=dc_i,
    search_all_decompose_dc=False)`` (api.cc:207-220), so rank ``r`` solves the candidates ``i % world == r`` on its own
    GPU.  The only exchange steps are the real ones: one all-reduce(MIN) over the candidate cost vector followed by the
    first-strict-minimum rule (api.cc:243-247), and a broadcast of the winning Pipeline from the rank that owns it.
    Every rank returns the same Pipeline, identical to the single-process ``solve``.  ``solver`` defaults to the HIP
    path; the CPU tests inject a stand-in."""
    import torch.distributed as dist

    rank, world, local, device = init()
    if solver is None:
        from . import _binary

        if _binary.device_count() > 0:
            _binary.set_device(local % _binary.device_count())
        solver = _binary.solve
    dcs, eff_hard_dc = candidate_list(int(kernel.shape[0]), hard_dc)
    mine = {}
    for i, dc in enumerate(dcs):
        if i % world == rank:  # interleaved: the expensive low-dc candidates land on different ranks
            mine[i] = solver(kernel, method0=method0, method1=method1, hard_dc=eff_hard_dc, decompose_dc=dc, qintervals=qintervals,
                             latencies=latencies, adder_size=adder_size, carry_size=carry_size, search_all_decompose_dc=False)  # fmt: skip
    import torch

    costs = torch.full((len(dcs),), float('inf'), dtype=torch.float32, device=device)
    for i, p in mine.items():
        costs[i] = pipeline_cost_f32(p)
    if world > 1:
        dist.all_reduce(costs, op=dist.ReduceOp.MIN)
    vals = costs.tolist()
    best = 0
    for i in range(1, len(vals)):
        if vals[i] < vals[best]:
            best = i
    if world == 1:
        return mine[best]
    box = [mine.get(best)]
    dist.broadcast_object_list(box, src=best % world)
    return box[0]


# ------------------------------------------------------------------------------------------------ column-sharded chains
class _DeviceView:
    """int32 view of ``count`` words of device memory owned by the HIP library, for ``torch.as_tensor`` (zero copy)."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {'shape': (int(count),), 'typestr': '<i4', 'data': (int(ptr), False), 'version': 2}


def make_allreduce_callback(device, abort=None, world=None):
    """C callback ``void(ctx, buf, count, on_device)`` = in-place all-reduce(sum) of int32 over the process group: the one
    collective a column-sharded chain uses (``csrc/cmvm_shard.h``).  RCCL (backend ``nccl``) works on device memory
    directly; with ``gloo`` (CPU tests, or several ranks sharing one GPU) device buffers are staged through the host.
    Returns (callback object -- keep it alive during the solve --, list collecting exceptions raised inside it).
    ``abort``: called after an exception inside the callback (``_binary.comm_abort``): the C side then stops at once -- an
    exception cannot unwind through it, and going on with a buffer that was not reduced would desynchronise the ranks."""
    import ctypes as C

    import numpy as np
    import torch
    import torch.distributed as dist

    FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)
    nccl = dist.is_initialized() and dist.get_backend() == 'nccl'
    errors: list[BaseException] = []

    def allreduce(ctx, buf, count, on_device):
        try:
            if world == 1 and not dist.is_initialized():
                return  # a single rank without a process group (the forced exchanges of a one-rank measurement): the sum over one rank
            if on_device and not torch.cuda.is_available() and os.environ.get('HIPEMU_DEVICES'):
                on_device = 0  # the emulated device of the CPU tests (tests/emu): its "device" memory is host memory
            if on_device:
                t = torch.as_tensor(_DeviceView(buf, count), device=device)
                if nccl:
                    dist.all_reduce(t)
                else:
                    h = t.cpu()
                    dist.all_reduce(h)
                    t.copy_(h)
                torch.cuda.synchronize(device)
            else:
                t = torch.from_numpy(np.ctypeslib.as_array((C.c_int32 * count).from_address(buf)))
                if nccl:
                    d = t.to(device)
                    dist.all_reduce(d)
                    t.copy_(d.cpu())
                else:
                    dist.all_reduce(t)
        except BaseException as e:  # an exception must not unwind through the C caller
            errors.append(e)
            if abort is not None:
                abort()

    return FN(allreduce), errors


def solve_column_sharded(kernel, method0: str = 'wmc', method1: str = 'auto', hard_dc: int = -1, decompose_dc: int = -2, qintervals=None,
                         latencies=None, adder_size: int = -1, carry_size: int = -1, search_all_decompose_dc: bool = True, sharded_solver=None,
                         return_stats: bool = False, transport: str = 'callback'):  # fmt: skip
    """One ``solve`` whose greedy chains are sharded over the output COLUMNS of their matrices (BASELINE config C4,
    SURVEY.md section 8e(2); ``csrc/cmvm_shard.h``): rank g holds the digits of columns [g n_out / W, (g+1) n_out / W), the
    pair table is replicated, every greedy step exchanges two all-reduce(sum) slabs (RCCL over xGMI on GPUs).  Every rank
    calls this with the same arguments and gets the same Pipeline, identical to the single-process ``solve``.

    Bound by the latency of ~2 collectives per greedy step (4 10^4 per 256x256 chain): it does not scale; the layout that
    does is ``solve_many_sharded``.  ``sharded_solver`` defaults to the HIP engine (``_binary.solve_sharded``); the CPU
    tests inject the sequential engine model.

    ``transport``: ``'callback'`` -- the collective is ``torch.distributed.all_reduce`` called back from the library (any
    backend; a host synchronisation per exchange); ``'rccl'`` -- the library's own RCCL transport (``csrc/cmvm_rccl.*``):
    ``ncclAllReduce`` in place on the library's stream, stream-ordered with its kernels, no Python in the loop; rank 0's
    RCCL unique id reaches the ranks through one ``torch.distributed.broadcast``."""
    rank, world, local, device = init()
    if transport == 'rccl':
        import numpy as np
        import torch
        import torch.distributed as dist

        from . import _binary

        if _binary.device_count() > 0:
            _binary.set_device(local % _binary.device_count())
        # ONE unique id per process group, broadcast once and re-used by every later solve: the library keeps the communicator
        # of an id (ncclCommInitRank costs tens of milliseconds and a rendezvous of all ranks), so a fresh id per call would
        # build and keep a new communicator each time.  Every rank takes the same branch: they all call in the same order.
        key = (world, dist.get_rank() if world > 1 else 0, str(device))
        raw = _rccl_ids.get(key)
        # Do all ranks still hold the group's id?  A rank whose previous solve raised has dropped it (below); the others may not have
        # failed at all (an argument error, a KeyboardInterrupt, an out-of-memory on one rank).  One tiny all-reduce(max) per solve --
        # not per greedy step -- makes the decision collective: if ANY rank needs an id, EVERY rank forgets its own (and the library's
        # communicators of it) and takes the fresh one of the broadcast; no rank can broadcast while the others skip it.
        need = 0 if raw is not None else 1
        if world > 1:
            need = int(max_over_ranks(need, device if dist.get_backend() == 'nccl' else None))
        if need:
            if raw is not None:  # (another rank lost its id: this rank's communicator of the old one goes too)
                _rccl_ids.clear()
                try:
                    _binary.rccl_shutdown()
                except Exception:
                    pass
            raw = _binary.rccl_unique_id() if rank == 0 else bytes(128)
            if world > 1:
                on_gpu = dist.get_backend() == 'nccl'
                t = torch.from_numpy(np.frombuffer(raw, np.uint8).copy())
                t = t.to(device) if on_gpu else t
                dist.broadcast(t, src=0)
                raw = t.cpu().numpy().tobytes()
            _rccl_ids[key] = raw
        try:
            pipe, stats = _binary.solve_sharded_rccl(kernel, raw, method0=method0, method1=method1, hard_dc=hard_dc, decompose_dc=decompose_dc, qintervals=qintervals,
                                                     latencies=latencies, adder_size=adder_size, carry_size=carry_size, search_all_decompose_dc=search_all_decompose_dc,
                                                     rank=rank, world=world)  # fmt: skip
        except BaseException:
            # a failed solve may leave the communicator of this id wedged: forget every cached id and the library's communicators (rccl_shutdown
            # destroys them all) on THIS rank; the agreement above makes the other ranks follow at the next call, whether they failed or not
            _rccl_ids.clear()
            try:
                _binary.rccl_shutdown()
            except Exception:
                pass
            raise
        return (pipe, stats) if return_stats else pipe
    if transport != 'callback':
        raise ValueError(f"transport must be 'callback' or 'rccl', not {transport!r}")
    if sharded_solver is None:
        from . import _binary

        if _binary.device_count() > 0:
            _binary.set_device(local % _binary.device_count())
        sharded_solver = _binary.solve_sharded
    abort = getattr(getattr(sharded_solver, '__self__', None), 'comm_abort', None)  # the engine model's, when the CPU tests inject it
    if abort is None:
        from . import _binary

        abort = _binary.comm_abort
    cb, errors = make_allreduce_callback(device, abort, world)
    try:
        pipe, stats = sharded_solver(kernel, method0=method0, method1=method1, hard_dc=hard_dc, decompose_dc=decompose_dc, qintervals=qintervals,
                                     latencies=latencies, adder_size=adder_size, carry_size=carry_size, search_all_decompose_dc=search_all_decompose_dc,
                                     rank=rank, world=world, allreduce=cb)  # fmt: skip
    except RuntimeError:
        if errors:  # the collective's own exception, not the library's "callback reported a failure"
            raise errors[0] from None
        raise
    if errors:
        raise errors[0]
    return (pipe, stats) if return_stats else pipe
