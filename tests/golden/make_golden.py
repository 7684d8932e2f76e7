"""Generate the golden fixtures of tests/golden/ from the REAL reference sources (oracle/_ref/libref.so, built by
oracle/Makefile from /root/reference against the container shim).  Run in the build container only:

    python tests/golden/make_golden.py

The fixtures pin the restated oracle (and through it the HIP path) to the reference itself; /root/reference is
never read by the tests."""
import gzip, hashlib, json, sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
sys.path.insert(0, str(HERE.parent))

import numpy as np  # noqa: E402
from cases import TEST_CMVM_GRID, int_matrix, random_case, reference_style_kernel  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def dump(p):
    """Pipeline -> plain JSON-able structure (same layout as Pipeline.save)"""
    return json.loads(json.dumps(p, default=lambda o: o.to_dict()))


def digest(p):
    return hashlib.sha256(json.dumps(dump(p), separators=(',', ':')).encode()).hexdigest()


def main():
    R = Oracle('ref')
    out = {'source': 'oracle/_ref/libref.so = /root/reference/src/da4ml/_binary/cmvm/*.cc @ reference snapshot 2026-05-15', 'full': [], 'digests': []}
    # full op lists (small cases)
    for seed in range(40):
        k, opts, _ = random_case(seed)
        out['full'].append({'case': f'random_case({seed})', 'result': dump(R.solve(k, **opts))})
    for seed in range(2):
        k = int_matrix(seed, 16, 16, -8, 8)
        for opts in (dict(), dict(adder_size=1, carry_size=-1)):
            out['full'].append({'case': f'c1 seed={seed} opts={sorted(opts.items())}', 'kernel': ['int_matrix', seed, 16, 16, -8, 8], 'opts': opts, 'result': dump(R.solve(k, **opts))})
    # digests + summary (larger / many cases)
    for n, bits in [(2, 2), (4, 4), (8, 8), (8, 2), (4, 8)]:
        k = reference_style_kernel(n * 10 + bits, n, bits)
        for gi, opts in enumerate(TEST_CMVM_GRID):
            p = R.solve(k, **opts)
            out['digests'].append({'case': f'grid n={n} bits={bits} #{gi}', 'kernel': ['reference_style_kernel', n * 10 + bits, n, bits], 'opts': opts, 'sha256': digest(p), 'cost': p.cost, 'n_ops': [len(s.ops) for s in p.solutions]})
    for seed in range(2, 8):
        k = int_matrix(seed, 16, 16, -8, 8)
        for opts in (dict(), dict(adder_size=1, carry_size=-1)):
            p = R.solve(k, **opts)
            out['digests'].append({'case': f'c1 seed={seed}', 'kernel': ['int_matrix', seed, 16, 16, -8, 8], 'opts': opts, 'sha256': digest(p), 'cost': p.cost, 'n_ops': [len(s.ops) for s in p.solutions]})
    for n, seed in [(32, 0), (48, 1)]:
        k = int_matrix(seed, n, n, -128, 128)
        for opts in (dict(), dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False), dict(method0='mc', hard_dc=2, adder_size=1, carry_size=-1)):
            p = R.solve(k, **opts)
            out['digests'].append({'case': f'int8 {n}x{n} seed={seed}', 'kernel': ['int_matrix', seed, n, n, -128, 128], 'opts': opts, 'sha256': digest(p), 'cost': p.cost, 'n_ops': [len(s.ops) for s in p.solutions]})
    k = int_matrix(0, 64, 64, -128, 128)
    opts = dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)
    p = R.solve(k, **opts)
    out['digests'].append({'case': 'c2 64x64 int8 seed=0 single chain', 'kernel': ['int_matrix', 0, 64, 64, -128, 128], 'opts': opts, 'sha256': digest(p), 'cost': p.cost, 'n_ops': [len(s.ops) for s in p.solutions]})
    with gzip.open(HERE / 'reference_golden.json.gz', 'wt') as f:
        json.dump(out, f, separators=(',', ':'))
    print('full', len(out['full']), 'digests', len(out['digests']))


if __name__ == '__main__':
    main()
