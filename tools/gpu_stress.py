"""Determinism stress: repeat the 256x256 default solve (21 chains, N = 8 and 9) and a 2-problem batch; all digests must agree."""
import sys, hashlib
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from cases import int_matrix
from da4ml_amd import _binary as hip
k = int_matrix(0, 256, 256, -128, 128)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seen = {}
for i in range(n):
    raw = hip.solve_many_raw([k, k] if i % 2 else [k])
    for j in range(len(raw)):
        s = raw.summary(j)
        L, h = hip.lib(), raw.handles[j]
        dig = hashlib.sha256()
        for st in range(L.da_n_stages(h)):
            info = np.zeros(5, np.int64); L.da_stage_info(h, st, info)
            a = [np.zeros(m, np.int64) for m in (int(info[0]), int(info[1]), int(info[1]), int(info[1]))]
            oi, of = np.zeros((int(info[2]), 4), np.int64), np.zeros((int(info[2]), 5), np.float32)
            L.da_stage_copy(h, st, *a, oi, of)
            for arr in (*a, oi, of): dig.update(arr.tobytes())
        key = dig.hexdigest()[:16]
        seen.setdefault(key, []).append((i, j, s['cost'], s['n_ops'], s['iterations'], int(L.da_picked(h))))
    raw.free()
for key, v in seen.items():
    print(key, len(v), v[0][2:], [x[:2] for x in v][:6])
print('DISTINCT RESULTS:', len(seen))
