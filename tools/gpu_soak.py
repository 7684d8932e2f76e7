"""Broad parity soak on the GPU box (not part of the test suite: minutes of oracle time): fresh random option sets, quantisation
steps that are not powers of two at small and medium sizes, every method at 40-56 square with the tracer's cost model, a 512x512
chain (functional criterion only: its oracle run would take most of a day).  Everything but the last against the CPU oracle.
usage: python tools/gpu_soak.py [n_random=2000]"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from cases import METHODS, int_matrix, odd_step_case, random_case
from da4ml_amd import _binary as hip
from oracle.oracle import Oracle

o = Oracle('port')
n_random = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
t0 = time.time()
bad = [s for s in range(5000, 5000 + n_random) if (lambda k, opts, _: hip.solve(k, **opts) != o.solve(k, **opts))(*random_case(s))]
print(f'random option sets 5000..{5000 + n_random - 1}: {len(bad)} mismatches {bad[:10]}  ({time.time() - t0:.0f} s)', flush=True)
t0 = time.time()
bad = [s for s in range(100, 700) if (lambda k, opts: hip.solve(k, **opts) != o.solve(k, **opts))(*odd_step_case(s))]
print(f'non-power-of-two steps, seeds 100..699: {len(bad)} mismatches {bad[:10]}  ({time.time() - t0:.0f} s)', flush=True)
t0 = time.time()
bad = []
rng = np.random.default_rng(77)
for i, method in enumerate(METHODS):
    for n in (40, 56):
        k = int_matrix(900 + 10 * i + n, n, n, -128, 128)
        st = (rng.choice(np.array([1.0, 3.0, 0.3, 1.7], np.float32), n) * 2.0 ** rng.integers(-3, 2, n)).astype(np.float32)
        q = [(float(-100 * s), float(90 * s), float(s)) for s in st]
        opts = dict(method0=method, method1='auto', hard_dc=2, decompose_dc=-2, adder_size=4, carry_size=8, qintervals=q, latencies=[float(v) for v in rng.integers(0, 3, n)],
                    search_all_decompose_dc=(n == 40))
        if hip.solve(k, **opts) != o.solve(k, **opts):
            bad.append((method, n))
print(f'every method at 40 / 56 square, cost model (4, 8), odd steps, hard_dc 2: {len(bad)} mismatches {bad}  ({time.time() - t0:.0f} s)', flush=True)
t0 = time.time()
k = int_matrix(512, 512, 512, -128, 128)
p = hip.solve(k, method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)
print(f'512x512 int8 chain: kernel reproduced {bool(np.all(p.kernel == k))}, {p.n_adders} adders  ({time.time() - t0:.0f} s)', flush=True)
