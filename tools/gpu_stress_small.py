"""Race hunt: solve small cases many times and compare with the oracle (computed once)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from cases import int_matrix
from da4ml_amd import _binary as hip
from oracle.oracle import Oracle
O = Oracle('port')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cases = [(int_matrix(s, 16, 16, -8, 8), dict(adder_size=1, carry_size=-1)) for s in range(4)] + [(int_matrix(7, 24, 24, -128, 128), dict())]
want = [O.solve(k, **o) for k, o in cases]
bad = 0; t0 = time.time()
for r in range(reps):
    got = hip.solve_many([k for k, _ in cases[:4]], adder_size=1, carry_size=-1) + [hip.solve(cases[4][0])]
    for i, (g, w) in enumerate(zip(got, want)):
        if g != w:
            bad += 1
            if bad <= 3:
                for si, (x, y) in enumerate(zip(g.solutions, w.solutions)):
                    if x != y:
                        d = next((j for j, (p, q) in enumerate(zip(x.ops, y.ops)) if p != q), None)
                        print(f'rep {r} case {i} stage {si}: n_ops {len(x.ops)} vs {len(y.ops)}, first op diff at {d}: {x.ops[d] if d is not None else None} vs {y.ops[d] if d is not None else None}', flush=True)
print(f'{reps} reps x {len(cases)} cases: mismatches {bad}  ({time.time()-t0:.1f}s)')
