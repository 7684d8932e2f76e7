"""Pass durations of k_iter_update's wavefronts by kind (a -DDA_UPD_TAIL build, DA4ML_HIP_LIB=...).  Usage: python tools/gpu_tail.py BATCH"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from cases import int_matrix
from da4ml_amd import _binary as hip
B = int(sys.argv[1])
ks = [int_matrix(s, 256, 256, -128, 128) for s in range(B)]
opts = dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)
hip.solve(ks[0][:8, :8].copy(), **opts)
hip.timings(reset=True)
raw = hip.solve_many_raw(ks, **opts); raw.free()
tm = hip.timings(reset=True)
allc, n, rc, rn, cn = tm['upd_fetch'], max(tm['upd_probe'], 1), tm['upd_cells'], tm['upd_blocks'], tm['upd_create']
cc, long20, long40, rgroups = tm['search_stale_rereads'], tm['search_touch_rereads'], tm['search_rounds'], tm['search_long_lists']
pn = max(n - rn - cn, 1)
print('passes %d: mean %.0f cycles | plain %d: %.0f | with creation %d: %.0f | with a rare case %d (%.2f %%, %.2f groups each): %.0f | > 20 k cycles %.2f %%, > 40 k %.3f %%; us/iter %.1f' % (
    n, allc / n, pn, (allc - rc - cc) / pn, cn, cc / max(cn, 1), rn, 100.0 * rn / n, rgroups / max(rn, 1), rc / max(rn, 1), 100.0 * long20 / n, 100.0 * long40 / n,
    1e3 * tm['loop_ms'] / max(tm['lockstep_iters'], 1)))
