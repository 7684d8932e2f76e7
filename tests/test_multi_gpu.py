"""world_size-2 gloo test of the multi-process sharding helpers (the N>1 path of bench.py / solve_many_sharded)."""

import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["DA_ROOT"])
from da4ml_amd import multi_gpu as mg
rank, world, local, device = mg.init("gloo")
assert world == 2 and device.type == "cpu"
lo, hi = mg.shard_bounds(7, rank, world)
mg.barrier()
t = mg.max_over_ranks(1.0 + rank)
s = mg.sum_over_ranks(hi - lo)
# candidate costs sharded over ranks: global first strict minimum
costs = [5.0, 3.0, 3.0, 9.0, 3.0, 8.0, 4.0]
best = mg.argmin_first(costs[lo:hi], lo, len(costs))
parts = mg.gather_to_rank0({"rank": rank, "shard": [lo, hi]})
if rank == 0:
    print(json.dumps({"t": t, "s": s, "best": best, "parts": parts}), flush=True)
mg.shutdown()
'''


def test_gloo_world2(tmp_path):
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), DA_ROOT=str(ROOT))
        procs.append(subprocess.Popen([sys.executable, '-c', WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e
    import json

    r = json.loads(outs[0][0].strip().splitlines()[-1])
    assert r['t'] == 2.0 and r['s'] == 7.0
    assert r['best'] == 1  # first strict minimum among equal costs
    assert r['parts'] == [{'rank': 0, 'shard': [0, 4]}, {'rank': 1, 'shard': [4, 7]}]


CAND_WORKER = r'''
import os, sys, json, hashlib
sys.path.insert(0, os.environ["DA_ROOT"]); sys.path.insert(0, os.path.join(os.environ["DA_ROOT"], "tests"))
from da4ml_amd import multi_gpu as mg
from oracle.oracle import Oracle   # stand-in solver: this host has no GPU
from cases import int_matrix
rank, world, local, device = mg.init("gloo")
O = Oracle("port")
out = []
for seed, kw in [(0, {}), (1, {"hard_dc": 2}), (2, {"adder_size": 1, "carry_size": -1})]:
    k = int_matrix(seed, 16, 16, -128, 128)
    p = mg.solve_candidates_sharded(k, solver=O.solve, **kw)
    out.append(bool(p == O.solve(k, **kw)))
print(json.dumps({"rank": rank, "same": out}), flush=True)
mg.shutdown()
'''


def test_candidate_sharded_solve_gloo_world2():
    """C4, candidate-sharded: the candidates of one searching solve split over two ranks give the single-process result
    on every rank (all-reduce(MIN) of the cost vector + broadcast of the winner)."""
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), DA_ROOT=str(ROOT))
        procs.append(subprocess.Popen([sys.executable, '-c', CAND_WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    import json

    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e
        r = json.loads(o.strip().splitlines()[-1])
        assert r['same'] == [True, True, True], r


def test_candidate_list_and_cost():
    from da4ml_amd import multi_gpu as mg
    from da4ml_amd.types import CombLogic, Op, Pipeline, QInterval

    assert mg.candidate_list(256) == (list(range(-1, 9)), 1_000_000_000)  # reference api.cc:190-201
    assert mg.candidate_list(16, 2) == ([-1, 0, 1, 2], 2)
    assert mg.candidate_list(1) == ([-1, 0], 1_000_000_000)
    q = QInterval(0.0, 1.0, 1.0)
    ops = [Op(0, -1, -1, 0, q, 0.0, 0.0)] + [Op(0, 0, 0, 0, q, 1.0, 16777216.0), Op(0, 0, 0, 0, q, 1.0, 1.0), Op(0, 0, 0, 0, q, 1.0, 1.0)]
    pipe = Pipeline((CombLogic((1, 1), [0], [3], [0], [False], ops, -1, -1),))
    assert mg.pipeline_cost_f32(pipe) == 16777216.0  # float32 accumulation in op order, like api.cc:222-229 (2**24 + 1 rounds back)


def test_shard_bounds_cover():
    from da4ml_amd.multi_gpu import shard_bounds

    for n in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_cost_estimate_and_balanced_shards():
    import numpy as np
    from cases import int_matrix

    from da4ml_amd.multi_gpu import balanced_shards, estimate_chain_cost

    # exact P_init of SURVEY.md section 8 for the C1 / C2 seed-0 matrices (+ n_in + n_out)
    assert estimate_chain_cost(int_matrix(0, 16, 16, -8, 8)) == 4119 + 32
    assert estimate_chain_cost(int_matrix(0, 64, 64, -128, 128)) == 1012155 + 128
    assert estimate_chain_cost(int_matrix(0, 16, 16, -8, 8) * 0.25) == 4119 + 32  # power-of-two scaling does not matter
    assert estimate_chain_cost(np.zeros((4, 4))) == 1.0 and estimate_chain_cost(np.zeros((0, 3))) == 0.0
    # a model-like batch: one big layer, many small ones
    costs = [900.0, 10.0, 10.0, 10.0, 300.0, 300.0, 300.0, 5.0]
    for world in (1, 2, 3, 8, 16):
        shards = balanced_shards(costs, world)
        assert sorted(i for s in shards for i in s) == list(range(len(costs))) and len(shards) == world
        assert all(s == sorted(s) for s in shards)
        assert shards == balanced_shards(costs, world)  # deterministic: every rank derives the same table
    two = balanced_shards(costs, 2)
    loads = [sum(costs[i] for i in s) for s in two]
    assert sorted(loads) == [915.0, 920.0]  # the contiguous split by count gives 930 / 905 here and 1210 / 625 for the rotated list
    assert balanced_shards([1.0] * 4, 2) == [[0, 2], [1, 3]]


MANY_WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["DA_ROOT"]); sys.path.insert(0, os.path.join(os.environ["DA_ROOT"], "tests"))
from da4ml_amd import multi_gpu as mg
from oracle.oracle import Oracle   # stand-in solver: this host has no GPU
from cases import int_matrix
rank, world, local, device = mg.init("gloo")
O = Oracle("port")
def solve_many(kernels, qintervals=None, latencies=None, **kw):
    return [O.solve(k, qintervals=None if qintervals is None else qintervals[i], latencies=None if latencies is None else latencies[i], **kw)
            for i, k in enumerate(kernels)]
shapes = [(12, 30), (3, 3), (4, 5), (20, 20), (2, 9), (6, 6), (16, 8)]
ks = [int_matrix(40 + i, a, b, -64, 64) for i, (a, b) in enumerate(shapes)]
lat = [[float(j % 3) for j in range(a)] for a, _ in shapes]
res = {}
for balance in ("cost", "count"):
    got = mg.solve_many_sharded(ks, balance=balance, solver_many=solve_many, latencies=lat, adder_size=1, carry_size=-1)
    if rank == 0:
        res[balance] = [bool(g == O.solve(k, latencies=l, adder_size=1, carry_size=-1)) for g, k, l in zip(got, ks, lat)]
    else:
        assert got is None
if rank == 0:
    print(json.dumps(res), flush=True)
mg.shutdown()
'''


def test_solve_many_sharded_gloo_world2():
    """C5: a batch of different layer shapes sharded over two ranks (by estimated cost and by count) comes back on rank 0
    in input order, every result equal to the single-process solve; per-matrix options follow their matrix."""
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), DA_ROOT=str(ROOT))
        procs.append(subprocess.Popen([sys.executable, '-c', MANY_WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    import json

    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e
    r = json.loads(outs[0][0].strip().splitlines()[-1])
    assert r == {'cost': [True] * 7, 'count': [True] * 7}, r


COLSHARD_WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["DA_ROOT"]); sys.path.insert(0, os.path.join(os.environ["DA_ROOT"], "tests"))
from da4ml_amd import multi_gpu as mg
from oracle.oracle import Oracle   # the sequential engine model stands in for the HIP engine: this host has no GPU
from cases import int_matrix, random_case
rank, world, local, device = mg.init("gloo")
M, O = Oracle("model"), Oracle("port")
out, steps = [], 0
cases = [(int_matrix(0, 16, 16, -128, 128), {}), (int_matrix(1, 24, 10, -128, 128), dict(method0="wmc", method1="wmc", decompose_dc=-1, search_all_decompose_dc=False)),
         (int_matrix(2, 9, 20, -8, 8), dict(adder_size=1, carry_size=-1)), (int_matrix(3, 12, 3, -64, 64), dict(hard_dc=1))]
cases += [random_case(s)[:2] for s in (1003, 1006, 1012, 1021)]
for k, kw in cases:
    p, st = mg.solve_column_sharded(k, sharded_solver=M.solve_sharded, return_stats=True, **kw)
    out.append(bool(p == O.solve(k, **kw)))
    steps += st["greedy_steps"]
    assert st["sharded_chains"] > 0 or k.shape[1] < world
print(json.dumps({"rank": rank, "same": out, "steps": steps}), flush=True)
mg.shutdown()
'''


def _run_world(worker, world, timeout=600, _retry=True):
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), DA_ROOT=str(ROOT))
        procs.append(subprocess.Popen([sys.executable, '-c', worker], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=timeout) for p in procs]
    if _retry and any(p.returncode != 0 for p in procs) and any(w in e for _, e in outs for w in ('Address already in use', 'Connection refused', 'Connection reset', 'timed out', 'Timed out')):
        return _run_world(worker, world, timeout, _retry=False)  # the free port was taken between probing and binding, or the rendezvous timed out on a loaded host: once more
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    import json

    def last_json(text):  # (a rank may print nothing of its own; gloo prints a connection banner on stdout)
        lines = [ln for ln in text.splitlines() if ln.startswith('{')]
        return json.loads(lines[-1]) if lines else None

    return [last_json(o) for o, _ in outs]


def test_column_sharded_chain_gloo_world2_and_3():
    """C4, column-sharded: every greedy chain of a solve split over the output columns of its matrix (pair table replicated,
    two all-reduce(sum) exchanges per greedy step, csrc/cmvm_shard.cc) gives the single-process result on EVERY rank --
    default search, single chain, tracer cost model, hard_dc, random option sets incl. custom intervals; 2 and 3 ranks
    (uneven column split; a 3-column matrix)."""
    for world in (2, 3):
        res = _run_world(COLSHARD_WORKER, world)
        assert [r['rank'] for r in res] == list(range(world))
        for r in res:
            assert all(r['same']) and len(r['same']) == 8, r
            assert r['steps'] > 100 and r['steps'] == res[0]['steps']  # the ranks walked the same greedy steps


COLSHARD8_WORKER = r'''
import os, sys, json, hashlib
sys.path.insert(0, os.environ["DA_ROOT"]); sys.path.insert(0, os.path.join(os.environ["DA_ROOT"], "tests"))
import torch.distributed as dist
from da4ml_amd import multi_gpu as mg
from oracle.oracle import Oracle   # the sequential engine model stands in for the HIP engine: this host has no GPU
from cases import int_matrix
rank, world, local, device = mg.init("gloo")
M = Oracle("model")
sizes = []                          # int32 elements of every exchange, in call order
plain_all_reduce = dist.all_reduce
def counting_all_reduce(t, *a, **kw):
    sizes.append(int(t.numel()))
    return plain_all_reduce(t, *a, **kw)
dist.all_reduce = counting_all_reduce
single = dict(method0="wmc", method1="wmc", decompose_dc=-1, search_all_decompose_dc=False)
res = {}
for name, k in (("c4_split_32x256", int_matrix(0, 32, 256, -128, 128)), ("uneven_20x45", int_matrix(1, 20, 45, -128, 128)), ("fewer_columns_than_ranks_12x5", int_matrix(2, 12, 5, -64, 64))):
    sizes.clear()
    p, st = mg.solve_column_sharded(k, sharded_solver=M.solve_sharded, return_stats=True, **single)
    dump = json.loads(json.dumps(p, default=lambda x: x.to_dict()))
    steps = st["greedy_steps"]
    # the exchanges of the greedy loop alternate: flags (+ status trailer), then the slab of partial count changes
    loop = sizes[1:1 + 2 * steps] if st["sharded_chains"] else []
    res[name] = {"sha": hashlib.sha256(json.dumps(dump, separators=(",", ":")).encode()).hexdigest(), "steps": steps, "chains": st["sharded_chains"], "calls": st["allreduce_calls"],
                 "flag_bytes_per_step": 4.0 * sum(loop[0::2]) / max(steps, 1), "slab_bytes_per_step": 4.0 * sum(loop[1::2]) / max(steps, 1), "reproduces": bool((p.kernel == k).all())}
print(json.dumps({"rank": rank, "res": res}), flush=True)
mg.shutdown()
'''


def test_column_sharded_chain_gloo_world8():
    """BASELINE config C4 at its OWN world size: 8 ranks, the 256 output columns of a 32 x 256 int8 matrix split 32 per rank (the
    config's split; 32 input rows keep the eight engine-model processes of this CPU test at a minute), an uneven split (45 columns
    over 8 ranks) and a matrix with fewer columns than ranks.  Every rank returns the
    same result -- digest equal on all eight and equal to the single-process oracle -- after the same greedy steps; the sizes of
    the two per-step exchanges are recorded (DESIGN.md section 7 projects the 8-GPU step time from them)."""
    import hashlib
    import json

    from cases import int_matrix
    from oracle.oracle import Oracle

    res = _run_world(COLSHARD8_WORKER, 8, timeout=1500)
    assert [r['rank'] for r in res] == list(range(8))
    single = dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)
    o = Oracle('port')
    for name, k in (('c4_split_32x256', int_matrix(0, 32, 256, -128, 128)), ('uneven_20x45', int_matrix(1, 20, 45, -128, 128)), ('fewer_columns_than_ranks_12x5', int_matrix(2, 12, 5, -64, 64))):
        want = json.loads(json.dumps(o.solve(k, **single), default=lambda x: x.to_dict()))
        sha = hashlib.sha256(json.dumps(want, separators=(',', ':')).encode()).hexdigest()
        for r in res:
            assert r['res'][name]['sha'] == sha and r['res'][name]['reproduces'], (name, r['rank'])
            assert r['res'][name]['steps'] == res[0]['res'][name]['steps'] and r['res'][name]['calls'] == res[0]['res'][name]['calls']
    c4 = res[0]['res']['c4_split_32x256']
    assert c4['chains'] >= 1 and c4['steps'] > 500 and c4['slab_bytes_per_step'] > c4['flag_bytes_per_step'] > 0
    print('C4 split at world 8:', json.dumps(c4))


def test_instance_and_candidate_sharding_gloo_world8():
    """the two layouts that have no per-step collective, at the size of a full node (8 ranks over gloo): the C5 batch of 7 layer
    shapes sharded by estimated cost and by count (7 units over 8 ranks: one rank gets nothing and still takes part in the gather)
    comes back on rank 0 in input order, equal to the single-process solves; the candidates of a searching solve split over 8
    ranks (10 candidates for a 16-column matrix: two ranks get two) give the single-process result on every rank."""
    res = _run_world(MANY_WORKER, 8, timeout=900)
    assert res[0] == {'cost': [True] * 7, 'count': [True] * 7}, res[0]
    res = _run_world(CAND_WORKER, 8, timeout=900)
    assert [r['rank'] for r in res] == list(range(8))
    for r in res:
        assert r['same'] == [True, True, True], r


def test_init_needs_a_port_from_the_launcher(monkeypatch):
    """Ranks started without a rendezvous port must fail loudly instead of guessing one."""
    from da4ml_amd import multi_gpu

    monkeypatch.setenv('RANK', '0')
    monkeypatch.setenv('WORLD_SIZE', '2')
    monkeypatch.setenv('LOCAL_RANK', '0')
    monkeypatch.delenv('MASTER_PORT', raising=False)
    with pytest.raises(RuntimeError, match='MASTER_PORT'):
        multi_gpu.init('gloo')


def test_failed_collective_stops_the_sharded_solve(model, monkeypatch):
    """A collective that raises inside the all-reduce callback (a peer died, a timeout) cannot unwind through the C code; the
    callback reports it (`comm_abort`) and the solve stops at that very exchange with the collective's own exception -- it used
    to go on with a buffer that was not reduced and raise only after the whole solve (if the peers ever let it finish)."""
    import numpy as np
    import torch.distributed as dist
    from cases import int_matrix

    from da4ml_amd import multi_gpu as mg

    calls = {'n': 0}

    def failing_all_reduce(t, *a, **kw):
        calls['n'] += 1
        if calls['n'] == 5:
            raise TimeoutError('simulated collective failure')

    monkeypatch.setattr(mg, 'init', lambda *a, **kw: (0, 2, 0, __import__('torch').device('cpu')))  # pretend to be rank 0 of 2
    monkeypatch.setattr(dist, 'all_reduce', failing_all_reduce)
    monkeypatch.setattr(dist, 'is_initialized', lambda: False)
    k = int_matrix(0, 16, 16, -128, 128)
    with pytest.raises(TimeoutError, match='simulated collective failure'):
        mg.solve_column_sharded(k, sharded_solver=model.solve_sharded, method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)
    assert calls['n'] == 5  # no further exchange after the failed one
    # the library is usable afterwards (one rank: no peer whose contribution the fake collective would have to supply)
    monkeypatch.setattr(mg, 'init', lambda *a, **kw: (0, 1, 0, __import__('torch').device('cpu')))
    p = mg.solve_column_sharded(k, sharded_solver=model.solve_sharded, method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)
    assert np.all(p.kernel == k)


def test_ranks_get_disjoint_core_slices(monkeypatch):
    """one process per GPU on one host: local rank r of n runs on its own slice of the allowed cores (the host pool of the
    library sizes itself from that mask), the slices are disjoint and cover the cores; DA4ML_PIN_RANKS=0 leaves the mask alone"""
    import os

    from da4ml_amd import multi_gpu as mg

    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip('one core')
    seen = []
    monkeypatch.setattr(mg, '_unpinned_cores', None)  # (module state of the idempotence: restored when the test ends)
    monkeypatch.setattr(mg, '_pinned_as', None)
    monkeypatch.setattr(os, 'sched_setaffinity', lambda pid, cores: seen.append(list(cores)))
    n = min(8, len(allowed))
    for r in range(n):
        mg.pin_rank_to_core_slice(r, n)
    assert sorted(c for s in seen for c in s) == allowed and all(len(s) >= 1 for s in seen)
    seen.clear()
    # idempotent (init() runs from every sharded solve): with a REAL narrowing of the mask in between, repeated calls return the same
    # slice of the ORIGINAL mask -- the slice is not sliced again
    monkeypatch.setattr(mg, '_unpinned_cores', None)
    monkeypatch.setattr(mg, '_pinned_as', None)
    current = list(allowed)
    monkeypatch.setattr(os, 'sched_getaffinity', lambda pid: set(current))

    def narrow(pid, cores):
        current[:] = list(cores)
        seen.append(list(cores))

    monkeypatch.setattr(os, 'sched_setaffinity', narrow)
    first = mg.pin_rank_to_core_slice(1, 2)
    assert first == allowed[len(allowed) // 2 :] and current == first
    assert mg.pin_rank_to_core_slice(1, 2) == first and mg.pin_rank_to_core_slice(1, 2) == first and len(seen) == 1
    assert mg.pin_rank_to_core_slice(0, 2) == allowed[: len(allowed) // 2]  # (other arguments: cut from the original mask again)
    monkeypatch.setattr(os, 'sched_setaffinity', lambda pid, cores: seen.append(list(cores)))
    monkeypatch.setattr(mg, '_unpinned_cores', None)
    monkeypatch.setattr(mg, '_pinned_as', None)
    seen.clear()
    assert mg.pin_rank_to_core_slice(0, 1) is None and not seen  # a single rank keeps every core
    monkeypatch.setenv('DA4ML_PIN_RANKS', '0')
    assert mg.pin_rank_to_core_slice(0, 4) is None and not seen


RCCL_ID_WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["DA_ROOT"])
import numpy as np
import torch.distributed as dist
from da4ml_amd import multi_gpu as mg
from da4ml_amd import _binary
rank, world, local, device = mg.init("gloo")
# the library's RCCL entry points replaced by stand-ins (no GPU here): what is under test is the PROTOCOL around them
made, shut, seen = [0], [0], []
def fake_id():
    made[0] += 1
    return bytes([made[0]]) * 128
def fake_shutdown():
    shut[0] += 1
    return 1
calls = [0]
def fake_solve(kernel, uid, rank=0, world=1, **kw):
    calls[0] += 1
    ids = [None] * world
    dist.all_gather_object(ids, uid)            # every rank must have been handed the SAME id
    assert len(set(ids)) == 1, ids
    seen.append(uid[0])
    return "pipe", {"allreduce_calls": 0}
_binary.rccl_unique_id, _binary.rccl_shutdown, _binary.solve_sharded_rccl = fake_id, fake_shutdown, fake_solve
_binary.device_count = lambda: 0
k = np.zeros((4, 4), np.float32)
mg.solve_column_sharded(k, transport="rccl")               # first solve: rank 0's id broadcast
mg.solve_column_sharded(k, transport="rccl")               # second: cached, no broadcast
first = list(seen)
# third solve: fails on rank 1 ONLY, inside the solve (behind the agreement, as an argument error or an allocation failure of one rank would);
# rank 0's solve ends normally (its stand-in skips the exchange this once): rank 1 forgets its id, rank 0 knows nothing of it
gather = [True]
def third(kernel, uid, rank=0, world=1, **kw):
    if rank == 1:
        raise ValueError("rank-local failure")
    return "pipe", {"allreduce_calls": 0}
_binary.solve_sharded_rccl = third
try:
    mg.solve_column_sharded(k, transport="rccl")
    failed = False
except ValueError:
    failed = True
assert failed == (rank == 1)
_binary.solve_sharded_rccl = fake_solve
mg.solve_column_sharded(k, transport="rccl")               # must not hang, must not mix ids: both ranks agree to take a fresh one
print(json.dumps({"rank": rank, "first": first, "last": seen[-1], "ids_made": made[0], "shutdowns": shut[0]}), flush=True)
mg.shutdown()
'''


def test_rccl_id_is_renewed_collectively_after_a_rank_local_failure():
    """a solve that fails on ONE rank made that rank broadcast for a fresh RCCL id at its next call while the others re-used the
    cached one and skipped the broadcast -- a collective mismatch.  Now one all-reduce(max) per solve decides together: any rank without an
    id makes every rank drop its own and take part in the broadcast.  (Transport stubbed: gloo, world 2, no GPU.)"""
    import json

    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), DA_ROOT=str(ROOT), DA4ML_PIN_RANKS='0')
        procs.append(subprocess.Popen([sys.executable, '-c', RCCL_ID_WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    r = [json.loads(o.strip().splitlines()[-1]) for o, _ in outs]
    assert r[0]['first'] == [1, 1] and r[1]['first'] == [1, 1]       # one id for the first two solves
    assert r[0]['last'] == r[1]['last'] == 2                          # a fresh one, the same on both ranks, afterwards
    assert r[0]['ids_made'] == 2 and r[0]['shutdowns'] >= 1 and r[1]['shutdowns'] >= 1
