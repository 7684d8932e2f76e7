"""Profiling workload: a batch of single greedy chains.  Usage: python tests/gpu_profile.py N BATCH"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from cases import int_matrix
from da4ml_amd import _binary as hip
n = int(sys.argv[1]); B = int(sys.argv[2])
ks = [int_matrix(s, n, n, -128, 128) for s in range(B)]
opts = dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)
hip.solve(ks[0][:8, :8].copy(), **opts)
hip.timings(reset=True)
t = time.time(); res = hip.solve_many(ks, _stats=True, **opts); dt = time.time() - t
tm = hip.timings(reset=True)
print(f'{n}x{n} batch {B}: {dt:.3f}s  -> {B/dt:.2f} solves/s; loop {tm["loop_ms"]:.1f} ms, lockstep iters {tm["lockstep_iters"]:.0f}, us/iter {1e3*tm["loop_ms"]/max(tm["lockstep_iters"],1):.1f}')
print(res[0][1], 'cost', res[0][0].cost, tm)
