"""The CPU oracle pinned against the golden vectors produced by the REAL reference sources (tests/golden/make_golden.py),
plus the reference's own property tests (reference tests/test_cmvm.py) run against the oracle."""

import gzip
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

import cases
from cases import TEST_CMVM_GRID, int_matrix, random_case, reference_style_kernel

GOLDEN = json.load(gzip.open(Path(__file__).parent / 'golden' / 'reference_golden.json.gz', 'rt'))


def dump(p):
    return json.loads(json.dumps(p, default=lambda o: o.to_dict()))


def digest(p):
    return hashlib.sha256(json.dumps(dump(p), separators=(',', ':')).encode()).hexdigest()


def golden_kernel(spec):
    return getattr(cases, spec[0])(*spec[1:])


def test_golden_full_op_lists(oracle):
    """op-for-op equality with the reference on the 40 random option sets and the C1 cases"""
    seen = 0
    for item in GOLDEN['full']:
        if item['case'].startswith('random_case('):
            k, opts, _ = random_case(int(item['case'][12:-1]))
        else:
            k, opts = golden_kernel(item['kernel']), item['opts']
        assert dump(oracle.solve(k, **opts)) == item['result'], item['case']
        seen += 1
    assert seen == 44


def test_golden_digests(oracle):
    """sha256 of the full result for the reference-test grid (5 kernels x 72 option sets), C1 seeds and int8 32/48/64"""
    for item in GOLDEN['digests']:
        n = item['kernel'][2]
        if n > 32:
            continue  # the larger ones run in test_golden_digests_large
        p = oracle.solve(golden_kernel(item['kernel']), **item['opts'])
        assert digest(p) == item['sha256'], item['case']
        assert p.cost == item['cost']


def test_golden_digests_large(oracle):
    for item in GOLDEN['digests']:
        if item['kernel'][2] <= 32:
            continue
        p = oracle.solve(golden_kernel(item['kernel']), **item['opts'])
        assert digest(p) == item['sha256'], item['case']
        assert [len(s.ops) for s in p.solutions] == item['n_ops']


def test_port_equals_reference_build(oracle):
    """Where oracle/_ref/libref.so exists (build container, and the GPU box via the snapshot) the restatement must equal
    the reference build on fresh random cases beyond the committed fixtures."""
    from oracle.oracle import HERE, Oracle

    if not (HERE / '_ref' / 'libref.so').exists():
        pytest.skip('oracle/_ref/libref.so not present')
    ref = Oracle('ref')
    for seed in range(1000, 1060):
        k, opts, _ = random_case(seed)
        assert oracle.solve(k, **opts) == ref.solve(k, **opts), seed
        assert all(np.array_equal(a, b) for a, b in zip(oracle.csd_decompose(k), ref.csd_decompose(k)))
        for dc in (-2, -1, 0, 1, 2):
            assert all(np.array_equal(a, b) for a, b in zip(oracle.kernel_decompose(k, dc), ref.kernel_decompose(k, dc)))
    from cases import odd_step_case

    for seed in range(40):  # quantisation steps that are not powers of two (state_opr.cc:57 takes log2 of any step)
        k, opts = odd_step_case(seed)
        assert oracle.solve(k, **opts) == ref.solve(k, **opts), seed


def test_large_records_of_the_restatement_match_the_reference_build():
    """tests/golden/large_*_golden.json: wherever a record of the reference's own build (key suffix _ref, made with
    oracle/_ref/libref.so) exists beside the restatement's record of the same problem, the two digests are identical --
    the restatement is pinned to the reference at 128x128 and at the benchmark's own 256x256 (seed 0: 49 minutes of libref, 72 of
    the restatement), not only up to 64x64.  (Seeds 4.. of the 256x256 chain have a reference-build record only.)"""
    import json
    from pathlib import Path

    pairs = []
    for name in ('large_chain_golden.json', 'large_default_golden.json'):
        gold = json.loads((Path(__file__).parent / 'golden' / name).read_text())
        for key, rec in gold.items():
            if key.endswith('_ref'):
                assert rec['oracle'] == 'oracle/_ref/libref.so'
                port = gold.get(key[:-4])
                if port is None:
                    continue
                assert (rec['sha256'], rec['cost'], rec['n_ops']) == (port['sha256'], port['cost'], port['n_ops']), key
                pairs.append(key)
    assert '128x128_seed0_single_chain_ref' in pairs and '256x256_seed0_single_chain_ref' in pairs
    assert '128x128_seed0_default_ref' in pairs and '256x256_seed0_default_ref' in pairs  # the full default search too (256x256: 96 minutes of libref on six threads, round 4)


# ---- the reference's own tests (tests/test_cmvm.py), seeded, against the oracle --------------------------------------
@pytest.mark.parametrize('n', [2, 4, 8])
@pytest.mark.parametrize('bits', [2, 4, 8])
def test_decompose_and_solve_properties(oracle, n, bits):
    k = reference_style_kernel(1000 + n * 10 + bits, n, bits)
    csd, s0, s1 = oracle.csd_decompose(k)
    rec = (csd * 2.0 ** s0[:, None, None] * 2.0 ** s1[None, :, None] * 2.0 ** np.arange(csd.shape[-1])[None, None, :]).sum(-1)
    assert np.all(rec == k)
    for dc in (-2, -1, 0, 1, 2):
        m0, m1 = oracle.kernel_decompose(k, dc)
        assert np.all(m0 @ m1 == k)
    for opts in TEST_CMVM_GRID[:: 3 if n == 8 else 1]:
        assert np.all(oracle.solve(k, **opts).kernel == k)


def test_unknown_method(oracle):
    with pytest.raises(RuntimeError, match='Unknown method'):
        oracle.solve(int_matrix(0, 8, 8, -8, 8), method0='nope', search_all_decompose_dc=False)
