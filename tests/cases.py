"""Seeded inputs shared by the CPU and GPU parity tests (no unseeded randomness, unlike the reference's tests)."""

import numpy as np


def int_matrix(seed, n_in, n_out, lo, hi):
    return np.random.default_rng(seed).integers(lo, hi, (n_in, n_out)).astype(np.float32)


def reference_style_kernel(seed, n, bits):
    """The generator of the reference's tests/test_cmvm.py:17-20, seeded."""
    r = np.random.default_rng(seed)
    return np.round((r.random((n, n)) - 0.5) * 2 ** (bits + 1)).astype(np.float32)


METHODS = ('mc', 'mc-dc', 'mc-pdc', 'wmc', 'wmc-dc', 'wmc-pdc')


def random_case(seed):
    """A random small matrix with a random option set, covering every method / cost-model / search combination."""
    rng = np.random.default_rng(seed)
    n_in, n_out = (int(v) for v in rng.integers(1, 13, 2))
    b = int(rng.integers(1, 10))
    k = (rng.random((n_in, n_out)).astype(np.float32) * 2**b - 2 ** (b - 1)).round()
    if seed % 5 == 0:
        k *= 2.0 ** int(rng.integers(-3, 3))
    if seed % 7 == 0:
        k[rng.integers(0, n_in)] = 0
    if seed % 11 == 0:
        k[:, rng.integers(0, n_out)] = 0
    k = np.ascontiguousarray(k, dtype=np.float32)
    opts = dict(
        method0=str(rng.choice(['mc', 'wmc', 'mc-dc', 'wmc-dc', 'mc-pdc', 'wmc-pdc'])),
        method1=str(rng.choice(['auto', 'mc', 'wmc', 'wmc-dc', 'mc-pdc'])),
        hard_dc=int(rng.choice([-1, 0, 1, 2, 3])),
        decompose_dc=int(rng.choice([-2, -1, 0, 1, 2])),
        adder_size=int(rng.choice([-1, 1, 4])),
        carry_size=int(rng.choice([-1, 2, 8])),
        search_all_decompose_dc=bool(rng.integers(0, 2)),
    )
    zero_input = False
    if seed % 3 == 0:
        lo = rng.integers(-64, 1, n_in)
        hi = lo + rng.integers(0, 200, n_in)
        st = 2.0 ** rng.integers(-3, 2, n_in)
        opts['qintervals'] = [(float(a * s), float(c * s), float(s)) for a, c, s in zip(lo, hi, st)]
        opts['latencies'] = [float(v) for v in rng.integers(0, 4, n_in)]
        if seed % 6 == 0:
            opts['qintervals'][0] = (0.0, 0.0, 1.0)
            zero_input = True
    return k, opts, zero_input


def odd_step_case(seed):
    """Input intervals whose quantisation steps are NOT powers of two (what the tracer's `variable * 3` produces, reference
    trace/fixed_variable.py:594): steps m * 2^k with up to five different mantissas per matrix, every cost model that looks at
    the steps, every method.  The latency model then needs -log2f(step) of the host libm for them (StepLog2, cmvm_core.h)."""
    rng = np.random.default_rng(50_000 + seed)
    n_in, n_out = (int(v) for v in rng.integers(2, 11, 2))
    k = rng.integers(-64, 64, (n_in, n_out)).astype(np.float32)
    mants = rng.choice(np.array([1.0, 3.0, 5.0, 0.3, 1.7, 6.25, 0.1], np.float32), size=int(rng.integers(1, 6)), replace=False)
    st = (rng.choice(mants, n_in) * 2.0 ** rng.integers(-4, 3, n_in)).astype(np.float32)
    lo = rng.integers(-40, 1, n_in)
    hi = lo + rng.integers(1, 120, n_in)
    opts = dict(
        method0=str(rng.choice(METHODS)),
        method1=str(rng.choice(['auto', 'wmc', 'mc-dc', 'wmc-pdc'])),
        hard_dc=int(rng.choice([-1, 0, 2])),
        decompose_dc=int(rng.choice([-2, -1, 0, 1])),
        adder_size=int(rng.choice([1, 4, -1])),
        carry_size=int(rng.choice([2, 8, -1])),
        search_all_decompose_dc=bool(rng.integers(0, 2)),
        qintervals=[(float(np.float32(a) * s), float(np.float32(c) * s), float(s)) for a, c, s in zip(lo, hi, st)],
        latencies=[float(v) for v in rng.integers(0, 3, n_in)],
    )
    if opts['adder_size'] < 0 and opts['carry_size'] < 0:
        opts['carry_size'] = 8  # at least one of the two, or the steps are never looked at
    return k, opts


TEST_CMVM_GRID = [
    dict(hard_dc=h, method0=m0, method1=m1, decompose_dc=d, search_all_decompose_dc=s, adder_size=1, carry_size=-1)
    for h in (0, 2, -1)
    for m0 in ('mc', 'wmc')
    for m1 in ('mc', 'wmc')
    for d in (0, -1, -2)
    for s in (False, True)
]


COST_MODELS = ((-1, -1), (1, -1), (4, 8))


def _method_grid():
    """(name, int_matrix arguments, options): every selector x hard_dc x cost model at 32x32 and 64x64 int8 with the default
    decomposition choice (search off: the `decompose_dc--` retry runs), and every selector as ONE 64x64 chain."""
    grid = []
    for n, seed in ((32, 5), (64, 6)):
        for m in METHODS:
            for h in (-1, 0, 2):
                for a, c in COST_MODELS:
                    opts = dict(method0=m, method1='auto', hard_dc=h, decompose_dc=-2, adder_size=a, carry_size=c, search_all_decompose_dc=False)
                    grid.append((f'{n}x{n} {m} hard_dc={h} adder={a} carry={c}', (seed, n, n, -128, 128), opts))
    for m in METHODS:
        for a, c in COST_MODELS:
            opts = dict(method0=m, method1=m, hard_dc=-1, decompose_dc=-1, adder_size=a, carry_size=c, search_all_decompose_dc=False)
            grid.append((f'64x64 single chain {m} adder={a} carry={c}', (7, 64, 64, -128, 128), opts))
    return grid


METHOD_GRID = _method_grid()
