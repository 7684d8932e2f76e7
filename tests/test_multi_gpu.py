"""world_size-2 gloo test of the multi-process sharding helpers (the N>1 path of bench.py / solve_many_sharded)."""

import os
import socket
import subprocess
import sys
from pathlib import Path

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
    print(json.dumps({"t": t, "s": s, "best": best, "parts": parts}))
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


def test_shard_bounds_cover():
    from da4ml_amd.multi_gpu import shard_bounds

    for n in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
