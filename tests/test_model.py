"""CPU proof that (1) the reformulated incremental algorithm of the GPU engine (tests/model/engine_model.cc, built from
the same cmvm_core.h inline functions as the kernels) picks exactly the reference's pairs and (2) the product's host
logic (da4ml_amd/csrc/cmvm_host.cc: option resolution, stage-1 MST, adder trees, candidate search) reproduces the
oracle's op lists.  No GPU involved; the GPU tests repeat the same comparisons through the HIP backend."""

import numpy as np
import pytest

from cases import TEST_CMVM_GRID, int_matrix, random_case, reference_style_kernel


@pytest.mark.parametrize('block', range(8))
def test_random_cases(model, oracle, block):
    for seed in range(block * 40, block * 40 + 40):
        k, opts, zero_input = random_case(seed)
        got = model.solve(k, **opts)
        assert got == oracle.solve(k, **opts), (seed, opts)
        if not zero_input:
            assert np.all(got.kernel == k)


def test_non_power_of_two_steps(model, oracle):
    """the engine model runs the same inline latency model as the kernels (cmvm_core.h) with the same host-built -log2f table"""
    from cases import odd_step_case

    for seed in range(40):
        k, opts = odd_step_case(seed)
        assert model.solve(k, **opts) == oracle.solve(k, **opts), seed


@pytest.mark.parametrize('n,bits', [(2, 2), (4, 4), (8, 8), (8, 4)])
def test_reference_grid(model, oracle, n, bits):
    k = reference_style_kernel(n * 10 + bits, n, bits)
    for opts in TEST_CMVM_GRID:
        assert model.solve(k, **opts) == oracle.solve(k, **opts), opts


def test_decompositions(model, oracle):
    for seed in range(30):
        k, _, _ = random_case(seed)
        assert all(np.array_equal(a, b) for a, b in zip(model.csd_decompose(k), oracle.csd_decompose(k)))
        assert all(np.array_equal(a, b) for a, b in zip(model.csd_decompose(k, center=False), oracle.csd_decompose(k, center=False)))
        for dc in (-2, -1, 0, 1, 2, 3):
            assert all(np.array_equal(a, b) for a, b in zip(model.kernel_decompose(k, dc), oracle.kernel_decompose(k, dc)))


def test_naf_equals_threshold_recoding(model, oracle):
    """the bit-trick NAF of cmvm_core.h against the reference's threshold recoding: every |x| < 2^13 and sampled large values"""
    x = np.arange(-(2**13), 2**13 + 1, dtype=np.int32)
    assert np.array_equal(model.int_arr_to_csd(x), oracle.int_arr_to_csd(x))
    rng = np.random.default_rng(5)
    for hi in (2**17, 2**24, 2**28):  # beyond 30 digits the reference overflows int32 (bit_decompose.cc:34-35)
        x = np.concatenate([rng.integers(-hi, hi, 4000), [hi - 1, -(hi - 1), hi // 3, hi // 3 + 1, 2 * hi // 3, 2 * hi // 3 + 1]]).astype(np.int32)
        assert np.array_equal(model.int_arr_to_csd(x), oracle.int_arr_to_csd(x))


@pytest.mark.parametrize('seed', range(4))
def test_c1_16x16_int4(model, oracle, seed):
    k = int_matrix(seed, 16, 16, -8, 8)
    for opts in (dict(), dict(adder_size=1, carry_size=-1)):
        assert model.solve(k, **opts) == oracle.solve(k, **opts)


def test_int8_32(model, oracle):
    k = int_matrix(0, 32, 32, -128, 128)
    for opts in (dict(), dict(method0='mc', method1='wmc', hard_dc=2, adder_size=1, carry_size=-1)):
        assert model.solve(k, **opts) == oracle.solve(k, **opts)


def test_wide_and_ragged(model, oracle):
    """> 16 CSD digits (64-bit cells), single row / column, all-zero rows and columns, fractional (shifted) entries"""
    ks = [
        int_matrix(3, 6, 5, -(2**19), 2**19),
        int_matrix(4, 1, 9, -64, 64),
        int_matrix(5, 9, 1, -64, 64),
        np.zeros((3, 4), np.float32),
        (int_matrix(6, 5, 5, -32, 32) * 2.0**-7).astype(np.float32),
    ]
    ks[1][0, 2] = 0
    for k in ks:
        for opts in (dict(), dict(search_all_decompose_dc=False, method0='mc-pdc', adder_size=2, carry_size=4)):
            assert model.solve(k, **opts) == oracle.solve(k, **opts)


def test_unknown_method(model):
    with pytest.raises(RuntimeError, match='Unknown method: nope'):
        model.solve(int_matrix(0, 8, 8, -8, 8), method0='nope', search_all_decompose_dc=False)
    # the reference only throws once the table is non-empty (cmvm_core.cc:36-64): a trivial matrix passes
    model.solve(np.eye(2, dtype=np.float32), method0='nope', method1='nope', search_all_decompose_dc=False)


def test_batch_memoises_identical_problems(oracle):
    """identical (matrix, intervals, latencies) problems of one batch are solved once; results are
    what the individual solves give, in input order, and only the unique problems reach the backend."""
    from oracle.oracle import Oracle

    model = Oracle('model')
    k0, k1 = int_matrix(0, 10, 8, -32, 32), int_matrix(1, 10, 8, -32, 32)
    q_a = [(-8.0, 7.0, 1.0)] * 10
    q_b = [(-4.0, 3.5, 0.5)] * 10
    kernels = [k0, k1, k0.copy(), k0, k1, k0]
    qints = [q_a, q_a, q_a, q_b, q_a, None]  # 0 == 2, 1 == 4; 3 and 5 differ from 0 in their intervals
    model.chains_run(reset=True)
    got = model.solve_many(kernels, qintervals=qints, adder_size=1, carry_size=-1)
    batch_chains = model.chains_run(reset=True)
    want = [oracle.solve(k, qintervals=q, adder_size=1, carry_size=-1) for k, q in zip(kernels, qints)]
    assert got == want
    uniq = [0, 1, 3, 5]
    model.solve_many([kernels[i] for i in uniq], qintervals=[qints[i] for i in uniq], adder_size=1, carry_size=-1)
    assert batch_chains == model.chains_run(reset=True)
    assert got[0] == got[2] and got[1] == got[4] and got[0] != got[3]


def test_default_search_record_64(model):
    """the 64x64 default-search record of tests/golden/large_default_golden.json through the product's host logic"""
    import hashlib
    import json
    from pathlib import Path

    rec = json.loads((Path(__file__).parent / 'golden' / 'large_default_golden.json').read_text())['64x64_seed0_default']
    p = model.solve(int_matrix(0, 64, 64, -128, 128))
    dump = json.loads(json.dumps(p, default=lambda o: o.to_dict()))
    assert p.cost == rec['cost'] and [len(s.ops) for s in p.solutions] == rec['n_ops']
    assert hashlib.sha256(json.dumps(dump, separators=(',', ':')).encode()).hexdigest() == rec['sha256']


def test_adder_trees_on_several_host_threads(model, oracle, monkeypatch):
    """finalize_chain reduces chunks of output columns on several host threads with chunk-local op ids; the op list must not
    depend on the number of threads / chunks (DA4ML_HIP_TREE_THREADS forces the chunked path on small chains)"""
    cases = [(int_matrix(3, 24, 40, -128, 128), dict(method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False)),
             (int_matrix(4, 32, 5, -8, 8), dict()), (int_matrix(5, 7, 33, -128, 128), dict(adder_size=1, carry_size=-1))]  # fmt: skip
    for k, opts in cases:
        want = oracle.solve(k, **opts)
        for threads in ('1', '2', '5', '64'):
            monkeypatch.setenv('DA4ML_HIP_TREE_THREADS', threads)
            assert model.solve(k, **opts) == want, (k.shape, opts, threads)
