import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from cases import int_matrix
from da4ml_amd import _binary as hip
B = int(sys.argv[1])
ks = [int_matrix(s, 256, 256, -128, 128) for s in range(B)]
opts = dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)
hip.solve(ks[0][:8, :8].copy(), **opts)
for r in range(2):
    t = time.time(); hip.solve_many_raw(ks, **opts).free(); print('call', r, 'batch', B, '%.3f s' % (time.time() - t), flush=True)
