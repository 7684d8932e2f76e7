"""Where a 64-chain C3 call spends its host time: input conversion, da_solve_batch, result release (GPU box).  Usage: python tools/host_split.py [reps]"""
import sys, time, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from cases import int_matrix
from da4ml_amd import _binary as hip
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ks = [int_matrix(s, 256, 256, -128, 128) for s in range(64)]
opts = dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)
hip.solve_many_raw(ks, **opts).free()
for r in range(reps):
    hip.timings(reset=True)
    t0 = time.perf_counter()
    kk = [hip._kernel(k) for k in ks]
    t1 = time.perf_counter()
    raw = hip.solve_many_raw(kk, **opts)
    t2 = time.perf_counter()
    raw.free()
    t3 = time.perf_counter()
    tm = hip.timings(reset=True)
    print('convert %.1f ms | solve_many_raw %.1f ms (library total %.1f, loop %.1f, col-dist %.1f) | free %.1f ms' % (
        1e3 * (t1 - t0), 1e3 * (t2 - t1), tm['total_ms'], tm['loop_ms'], tm['dist_ms'], 1e3 * (t3 - t2)))
