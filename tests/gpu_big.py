import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from cases import int_matrix
from da4ml_amd import _binary as hip
for shape in [(512, 512), (384, 1024)]:
    k = int_matrix(1, *shape, -128, 128)
    hip.timings(reset=True)
    t = time.time(); p = hip.solve(k, method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False); dt = time.time() - t
    tm = hip.timings()
    ok = bool(np.all(p.kernel == k))
    print(shape, 'solve %.2fs' % dt, 'cost', p.cost, 'ops', [len(s.ops) for s in p.solutions], 'kernel reproduced:', ok, 'iters', tm['iterations'], 'retries', tm['retries'], 'arena GB %.2f' % (tm['arena_bytes']/1e9), flush=True)
