"""Worker of tests/test_emulated_device.py: runs in a process whose DA4ML_HIP_LIB points at tests/emu/libda4ml_emu.so, i.e.
the product's Python layer, C ABI, host logic AND kernels, the latter executed thread by thread on the CPU.  Everything is
compared with the oracle; one JSON line on stdout.  usage: worker.py <what> [args]"""

import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))

import numpy as np  # noqa: E402
from cases import int_matrix, random_case  # noqa: E402

from da4ml_amd import _binary as hip  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

SINGLE = dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)


def out(**kw):
    print(json.dumps(kw), flush=True)


def random_cases(lo, hi):
    o = Oracle('port')
    bad = []
    for seed in range(lo, hi):
        k, opts, _ = random_case(seed)
        if hip.solve(k, **opts) != o.solve(k, **opts):
            bad.append(seed)
    out(bad=bad, n=hi - lo)


def odd_steps(lo, hi):
    """input steps that are not powers of two: the table-driven -log2f of the latency model (StepLog2)"""
    from cases import odd_step_case

    o = Oracle('port')
    bad = []
    for seed in range(lo, hi):
        k, opts = odd_step_case(seed)
        if hip.solve(k, **opts) != o.solve(k, **opts):
            bad.append(seed)
    # 40 different non-power-of-two step mantissas in one matrix: one table row each (the reference takes -log2 of any step)
    k = np.random.default_rng(5).integers(-16, 16, (40, 6)).astype(np.float32)
    q = [(-8.0 * (1.0 + 0.017 * (i + 1)), 8.0 * (1.0 + 0.017 * (i + 1)), 1.0 + 0.017 * (i + 1)) for i in range(40)]
    for j, opts in enumerate((dict(adder_size=1, carry_size=-1), dict(adder_size=4, carry_size=8), {})):
        if hip.solve(k, qintervals=q, **opts) != o.solve(k, qintervals=q, **opts):
            bad.append(1000 + j)
    out(bad=bad, n=hi - lo)


def layouts():
    """both entry layouts and their boundaries: narrow u32 entries (<= 256 columns, <= 12 digits) and the wide 16-byte ones"""
    o = Oracle('port')
    cases = {
        'narrow_12_digits': (int_matrix(1, 6, 5, -2048, 2048), SINGLE),
        'wide_13_digits': (int_matrix(2, 6, 5, -4096, 4096), SINGLE),
        'wide_fractional': (int_matrix(3, 5, 6, -128, 128).astype(np.float32) / 64.0, SINGLE),
        'columns_256': (int_matrix(4, 3, 256, -2, 2), SINGLE),
        'columns_257': (int_matrix(5, 3, 257, -2, 2), SINGLE),
        'one_row': (int_matrix(6, 1, 20, -128, 128), SINGLE),
        'one_column': (int_matrix(7, 20, 1, -128, 128), SINGLE),
        'zeros': (np.zeros((4, 4), np.float32), SINGLE),
        'default_search': (int_matrix(8, 12, 12, -32, 32), {}),
    }
    bad = [name for name, (k, opts) in cases.items() if hip.solve(k, **opts) != o.solve(k, **opts)]
    out(bad=bad, n=len(cases))


def batch():
    """many chains per launch, repeated problems, mixed shapes (both layouts in one batch)"""
    o = Oracle('port')
    ks = [int_matrix(s, 6 + s % 5, 4 + s % 7, -64, 64) for s in range(12)] + [int_matrix(3, 9, 7, -64, 64), int_matrix(40, 4, 5, -8192, 8192)]
    got = hip.solve_many(ks, **SINGLE)
    bad = [i for i, k in enumerate(ks) if got[i] != o.solve(k, **SINGLE)]
    tm = hip.timings()
    out(bad=bad, n=len(ks), chains=tm['chains'])


def big_table():
    """a pair table of 2 M slots (environment set by the test) for small problems: 4096 groups of 512 slots, i.e. the geometry
    of a 256x256 chain -- every lane of k_iter_select holds four group bounds and re-reads eight slots per group, which the
    tables of small problems (4 groups of 256 slots) never reach"""
    o = Oracle('port')
    ks = [int_matrix(11, 14, 9, -128, 128), int_matrix(12, 9, 14, -4096, 4096)]
    got = hip.solve_many(ks, **SINGLE)
    bad = [i for i, k in enumerate(ks) if got[i] != o.solve(k, **SINGLE)]
    k = int_matrix(13, 10, 10, -32, 32)
    if hip.solve(k, method0='mc-dc', method1='wmc-pdc', adder_size=1, carry_size=-1) != o.solve(k, method0='mc-dc', method1='wmc-pdc', adder_size=1, carry_size=-1):
        bad.append(2)
    out(bad=bad)


def record(name, fname):
    """a committed large record (tests/golden/large_*_golden.json: 128x128 / 256x256 chains and default searches, made by the
    oracle or by oracle/_ref/libref.so in up to 141 minutes of CPU) reproduced by the kernels on the emulated device"""
    import hashlib
    import re

    rec = json.loads((ROOT / 'tests' / 'golden' / fname).read_text())[name]
    n, seed = (int(v) for v in re.match(r'(\d+)x\1_seed(\d+)_', name).groups())
    p = hip.solve(int_matrix(seed, n, n, -128, 128), **rec['opts'])
    dump = json.loads(json.dumps(p, default=lambda o: o.to_dict()))
    sha = hashlib.sha256(json.dumps(dump, separators=(',', ':')).encode()).hexdigest()
    out(equal=p.cost == rec['cost'] and [len(s.ops) for s in p.solutions] == rec['n_ops'] and sha == rec['sha256'], cost=p.cost)


def fork_after_use():
    """the host thread pool of the library (parked threads, cmvm_host.cc) does not exist in a forked child: the child must start
    its own instead of waiting for threads that were never copied"""
    ks = [int_matrix(s, 16, 16, -128, 128) for s in range(8)]
    a = hip.solve_many(ks, **SINGLE)
    pid = os.fork()
    if pid == 0:
        os._exit(0 if hip.solve_many(ks, **SINGLE) == a else 3)
    _, st = os.waitpid(pid, 0)
    # fork() while ANOTHER thread is inside a solve returns at once (the library takes no lock around the fork: a prepare handler that
    # waited for the solve-long library lock would stall here for the whole solve -- holding the GIL).  Measured against the duration of
    # the solve itself (both scale with the load of the host), and the busy thread says when it has started.
    import threading
    import time

    big = [int_matrix(s, 28, 28, -128, 128) for s in range(3)]
    t0 = time.time()
    hip.solve_many(big, **SINGLE)
    t_solve = time.time() - t0
    started = threading.Event()

    def work():
        started.set()
        for _ in range(3):
            hip.solve_many(big, **SINGLE)

    busy = threading.Thread(target=work)
    busy.start()
    started.wait()
    time.sleep(min(0.3, 0.25 * t_solve))
    t0 = time.time()
    pid2 = os.fork()
    if pid2 == 0:
        # the child of a fork taken in mid-solve: the library says that its copied state is unusable, instead of touching it
        try:
            hip.solve_many(ks, **SINGLE)
            os._exit(4)
        except RuntimeError as e:
            os._exit(0 if 'forked while another thread' in str(e) else 5)
    fork_seconds = time.time() - t0
    in_solve = busy.is_alive()
    _, st2 = os.waitpid(pid2, 0)
    busy.join()
    out(child=os.WEXITSTATUS(st), parent=hip.solve_many(ks, **SINGLE) == a, fork_during_solve_ok=bool(in_solve and fork_seconds < max(0.5, 1.5 * t_solve)),
        child_of_busy_fork=os.WEXITSTATUS(st2), fork_seconds=fork_seconds, solve_seconds=t_solve)


def lds_budget():
    """HIPEMU_STATIC_LDS makes the emulated kernels report static __shared__ bytes: the product's budget check (static + dynamic against the
    device's per-workgroup limit) then drops the optional claim area first -- same results -- and refuses with a clear message beyond that"""
    o = Oracle('port')
    ks = [int_matrix(s, 10 + s, 12, -64, 64) for s in range(4)]
    try:
        got = hip.solve_many(ks, **SINGLE)
        out(ok=True, bad=[i for i, k in enumerate(ks) if got[i] != o.solve(k, **SINGLE)], message='')
    except RuntimeError as e:
        out(ok=False, bad=[], message=str(e))


def retry():
    """arena heuristics far too small (environment set by the test): capacity error on the device, rerun with larger arenas"""
    o = Oracle('port')
    k = int_matrix(0, 20, 20, -128, 128)
    p = hip.solve(k, **SINGLE)
    out(equal=p == o.solve(k, **SINGLE), retries=hip.timings()['retries'])


def shard_single():
    """the column-sharded engine (k_cs_* kernels, k_iter_select<SHARDED>) with one rank: exchanges are no-ops"""
    o = Oracle('port')
    bad = []
    for seed, shape in ((1, (10, 12)), (2, (7, 5)), (3, (4, 300))):
        k = int_matrix(seed, *shape, -16, 16)
        p, st = hip.solve_sharded(k, **SINGLE)
        if p != o.solve(k, **SINGLE) or st['sharded_chains'] < 1:
            bad.append(seed)
    out(bad=bad)


def shard_retry():
    """column-sharded chain with arena heuristics far too small: the capacity error travels in the status trailer of the flag
    exchange, every rank reruns with four times the capacities (ShardedBackend::run_one) -- the result is the oracle's"""
    o = Oracle('port')
    k = int_matrix(0, 20, 20, -128, 128)
    p, st = hip.solve_sharded(k, **SINGLE)
    out(equal=p == o.solve(k, **SINGLE), chains=st['sharded_chains'], steps=st['greedy_steps'], calls=st['allreduce_calls'])


def shard_rank():
    """one rank of a gloo job: the HIP shard engine of every rank runs on its own emulated device"""
    import ctypes as C
    import hashlib

    import torch
    import torch.distributed as dist

    from da4ml_amd import multi_gpu as mg

    rank, world, _, _ = mg.init('gloo')
    FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)

    def allreduce(ctx, buf, count, on_device):  # the emulated device's memory is host memory
        t = torch.from_numpy(np.ctypeslib.as_array((C.c_int32 * count).from_address(buf)))
        dist.all_reduce(t)

    cb = FN(allreduce)

    def solver(kernel, allreduce=None, **kw):
        return hip.solve_sharded(kernel, allreduce=cb, **kw)

    digests = []
    for seed, shape, opts in ((1, (10, 12), SINGLE), (2, (9, 7), {}), (3, (6, 5), dict(adder_size=1, carry_size=-1))):
        k = int_matrix(seed, *shape, -16, 16)
        p = mg.solve_column_sharded(k, sharded_solver=solver, **opts)
        dump = json.dumps(json.loads(json.dumps(p, default=lambda x: x.to_dict())), separators=(',', ':'))
        digests.append(hashlib.sha256(dump.encode()).hexdigest())
    if rank == 0:
        o = Oracle('port')
        want = []
        for seed, shape, opts in ((1, (10, 12), SINGLE), (2, (9, 7), {}), (3, (6, 5), dict(adder_size=1, carry_size=-1))):
            p = o.solve(int_matrix(seed, *shape, -16, 16), **opts)
            want.append(hashlib.sha256(json.dumps(json.loads(json.dumps(p, default=lambda x: x.to_dict())), separators=(',', ':')).encode()).hexdigest())
        Path(os.environ['EMU_OUT']).write_text(json.dumps({'equal': digests == want}))
    gathered = [None] * world
    dist.all_gather_object(gathered, digests)
    assert all(g == digests for g in gathered)
    mg.shutdown()


def race_cases():
    """workload of the race-detector run (no oracle in this process: its OpenMP runtime is not instrumented); the digests are
    compared by the caller.  Several partner rows per update block, two chains per launch, both entry layouts, an insert-heavy
    default search"""
    import hashlib

    def digest(p):
        return hashlib.sha256(json.dumps(json.loads(json.dumps(p, default=lambda x: x.to_dict())), separators=(',', ':')).encode()).hexdigest()

    ks = [int_matrix(1, 28, 5, -128, 128), int_matrix(3, 5, 4, -4096, 4096)]
    got = hip.solve_many(ks, **SINGLE)
    tm = hip.timings()
    out(digests=[digest(p) for p in got] + [digest(hip.solve(int_matrix(4, 8, 8, -32, 32)))], partners_per_step=tm['partners'] / max(tm['iterations'], 1))


def dais():
    """k_dais_run on the emulated device against the host executor"""
    from dais_cases import random_program

    bad = []
    for seed in range(12):
        prog, x = random_program(seed, n_samples=70)  # more than one wavefront of samples
        if not np.array_equal(hip.dais_interp_run(prog, x, executor='device'), hip.dais_interp_run(prog, x, executor='host')):
            bad.append(seed)
    out(bad=bad)


if __name__ == '__main__':
    what = sys.argv[1]
    {'random': lambda: random_cases(int(sys.argv[2]), int(sys.argv[3])), 'oddsteps': lambda: odd_steps(int(sys.argv[2]), int(sys.argv[3])), 'layouts': layouts, 'batch': batch, 'lds_budget': lds_budget, 'retry': retry, 'big_table': big_table, 'fork': fork_after_use, 'record': lambda: record(sys.argv[2], sys.argv[3]),
     'shard_single': shard_single, 'shard_retry': shard_retry, 'shard_rank': shard_rank, 'dais': dais, 'race_cases': race_cases}[what]()  # fmt: skip
