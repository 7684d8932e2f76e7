"""Time of ONE column-sharded chain on one GPU (sharded phases forced, the exchanges are no-ops) next to the ordinary
single-GPU chain: the per-step cost of the sharded path itself (3 host synchronisations + 2 small read-backs per greedy
step), i.e. what every rank pays before any xGMI latency; then the same with the library's RCCL transport in the loop (an all-reduce over
one rank for every exchange: stream-ordered ncclAllReduce, no Python).  usage: python tools/shard_bench.py [n=256]"""
import os, sys, time
os.environ['DA4ML_SHARD_FORCE'] = '1'
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from da4ml_amd import _binary as hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
k = np.random.default_rng(0).integers(-128, 128, (n, n)).astype(np.float32)
opts = dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)
hip.solve(k[:8, :8].copy(), **opts)
t = time.time(); p0 = hip.solve(k, **opts); t_plain = time.time() - t
t = time.time(); p1, st = hip.solve_sharded(k, rank=0, world=1, **opts); t_shard = time.time() - t
print(f'{n}x{n} single chain: ordinary {t_plain:.3f} s; column-sharded path (1 rank, exchanges no-ops) {t_shard:.3f} s for {st["greedy_steps"]} greedy steps '
      f'= {1e6 * t_shard / max(st["greedy_steps"], 1):.1f} us per step, {st["allreduce_calls"]} all-reduce calls; results identical: {p0 == p1}; adders {p1.n_adders}')
os.environ['DA4ML_SHARD_FORCE_COMM'] = '1'
try:
    uid = hip.rccl_unique_id()
    hip.solve_sharded_rccl(k[:16, :16].copy(), uid, rank=0, world=1, **opts)  # communicator set-up outside the timing
    t = time.time(); p2, st2 = hip.solve_sharded_rccl(k, uid, rank=0, world=1, **opts); t_rccl = time.time() - t
    print(f'{n}x{n} single chain: column-sharded path with the RCCL transport in the loop (1 rank, ncclAllReduce per exchange) {t_rccl:.3f} s '
          f'= {1e6 * t_rccl / max(st2["greedy_steps"], 1):.1f} us per step, {st2["allreduce_calls"]} all-reduce calls; results identical: {p0 == p2}')
except Exception as e:  # RCCL absent or unusable on this box
    print(f'RCCL transport not measured: {type(e).__name__}: {e}')
