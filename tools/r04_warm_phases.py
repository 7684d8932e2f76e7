import sys, time, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from cases import int_matrix
from da4ml_amd import _binary as hip
n, B = 256, 64
ks = [int_matrix(s, n, n, -128, 128) for s in range(B)]
opts = dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)
for rep in range(3):
    if rep == 2: os.environ['DA4ML_HIP_VERBOSE'] = '1'
    t = time.time(); raw = hip.solve_many_raw(ks, **opts); dt = time.time() - t; raw.free()
    print(f'call {rep}: {dt*1e3:.1f} ms', flush=True)
