"""Ad-hoc GPU bring-up script (not a pytest file): runs increasingly large cases against the oracle and reports the
first divergence.  Usage on the GPU box: python tests/gpu_debug.py [max_cases]"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from cases import random_case, int_matrix
from da4ml_amd import _binary as hip
from oracle.oracle import Oracle
O = Oracle('port')
print('devices', hip.device_count(), flush=True)
def first_diff(a, b):
    for si, (x, y) in enumerate(zip(a.solutions, b.solutions)):
        if x == y: continue
        for f in ('shape', 'inp_shifts', 'out_idxs', 'out_shifts', 'out_negs', 'carry_size', 'adder_size'):
            if getattr(x, f) != getattr(y, f): return f'stage {si} field {f}: {getattr(x, f)} vs {getattr(y, f)}'
        if len(x.ops) != len(y.ops): 
            n = min(len(x.ops), len(y.ops))
            for i in range(n):
                if x.ops[i] != y.ops[i]: return f'stage {si} n_ops {len(x.ops)} vs {len(y.ops)}; first op diff at {i}: {x.ops[i]} vs {y.ops[i]}'
            return f'stage {si} n_ops {len(x.ops)} vs {len(y.ops)} (common prefix equal)'
        for i, (p, q) in enumerate(zip(x.ops, y.ops)):
            if p != q: return f'stage {si} op {i}: {p} vs {q}'
    return 'equal'
bad = 0
k = int_matrix(0, 4, 4, -8, 8)
for opts in (dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False), dict()):
    t = time.time(); g = hip.solve(k, **opts); dt = time.time() - t
    w = O.solve(k, **opts)
    print('4x4', opts, 'equal' if g == w else first_diff(g, w), '%.3fs' % dt, flush=True)
    bad += g != w
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for seed in range(N):
    k, opts, z = random_case(seed)
    try:
        g = hip.solve(k, **opts)
    except Exception as e:
        print('seed', seed, 'EXC', e, opts, flush=True); bad += 1; continue
    w = O.solve(k, **opts)
    if g != w:
        bad += 1
        print('seed', seed, k.shape, opts, first_diff(g, w), flush=True)
print('random small: bad =', bad, 'of', N, flush=True)
for n in (16, 32, 64):
    k = int_matrix(0, n, n, -128, 128)
    opts = dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)
    t = time.time(); g, st = hip.solve(k, _stats=True, **opts); dt = time.time() - t
    w = O.solve(k, **opts)
    print(n, 'single chain', 'equal' if g == w else first_diff(g, w), '%.3fs' % dt, st, hip.timings(reset=True), flush=True)
    t = time.time(); g, st = hip.solve(k, _stats=True); dt = time.time() - t
    w = O.solve(k)
    print(n, 'default', 'equal' if g == w else first_diff(g, w), '%.3fs' % dt, st, hip.timings(reset=True), flush=True)
