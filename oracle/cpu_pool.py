"""Process-pool timing of the CPU reference path on all host cores -- TEST / BENCHMARK INFRASTRUCTURE ONLY
(used by ``bench.py``'s ``cpu_baseline`` leg; never imported by the product).

The reference parallelises one ``solve`` only over its <= 10 ``decompose_dc`` candidates (api.cc:208) and callers invoke
it sequentially, so the CPU's best case for a batch of INDEPENDENT matrices is one process per core, each running the
single-threaded solver on its own matrix.  Workers are started with the ``spawn`` method: they
import numpy and the oracle's ctypes front-end only, never the HIP library.
"""

from __future__ import annotations

import os
import time


def host_cores() -> int:
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        return os.cpu_count() or 1


def mem_available_gb() -> float:
    try:
        for line in open('/proc/meminfo'):
            if line.startswith('MemAvailable:'):
                return int(line.split()[1]) / 1048576.0
    except OSError:
        pass
    return 16.0


def _matrix(n_in, n_out, seed):
    import numpy as np

    return np.random.default_rng(seed).integers(-128, 128, (n_in, n_out)).astype(np.float32)


def _oracle(kind):
    os.environ['OMP_NUM_THREADS'] = '1'
    from oracle.oracle import Oracle

    return Oracle(kind)


def sample_worker(job):
    """time-bounded prefix of one greedy chain: (kind, n_in, n_out, seed, method, budget_s) -> dict"""
    kind, n_in, n_out, seed, method, budget_s = job
    from oracle.oracle import sample_chain

    o = _oracle(kind)
    t = time.perf_counter()
    s = sample_chain(o, _matrix(n_in, n_out, seed), method, budget_s)
    s['wall_s'] = time.perf_counter() - t
    s['seed'] = seed
    return s


def solve_worker(job):
    """one complete solve: (kind, n_in, n_out, seed, opts) -> (seconds inside the C call, cost, n_ops per stage)"""
    kind, n_in, n_out, seed, opts = job
    o = _oracle(kind)
    k = _matrix(n_in, n_out, seed)
    t = time.perf_counter()
    p = o.solve(k, **opts)
    dt = time.perf_counter() - t
    return dt, p.cost, [len(s.ops) for s in p.solutions]


def problem_worker(job):
    """one complete solve of an explicit problem: (kind, kernel, opts incl. qintervals / latencies) -> (seconds, cost)"""
    kind, kernel, opts = job
    o = _oracle(kind)
    t = time.perf_counter()
    p = o.solve(kernel, **opts)
    return time.perf_counter() - t, p.cost


def run_pool(fn, jobs, workers):
    """map ``fn`` over ``jobs`` on ``workers`` spawned processes; returns (results in job order, wall seconds of the map)"""
    import multiprocessing as mp

    # the workers are single-threaded solvers: keep the numerical libraries they import from starting one thread per core each
    # (64 processes x 256 BLAS threads survive on the 256-core box, 256 x 256 did not)
    for var in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS', 'NUMEXPR_NUM_THREADS'):
        os.environ.setdefault(var, '1')
    ctx = mp.get_context('spawn')
    with ctx.Pool(processes=workers) as pool:
        pool.map(_warm, range(workers))  # interpreter start-up and imports are not part of the timed map
        t = time.perf_counter()
        out = pool.map(fn, jobs, chunksize=1)
        wall = time.perf_counter() - t
    return out, wall


def _warm(_):
    import numpy  # noqa: F401

    import oracle.oracle  # noqa: F401

    return os.getpid()
