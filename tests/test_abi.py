"""The C-ABI library loads on a GPU-less host, exports every symbol declared in include/da4ml_hip.h, serves the pure
scalar helpers, and fails loudly (no CPU fallback) for everything that needs the device."""

import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope='module')
def hip():
    import __graft_entry__ as entry

    if not (ROOT / 'da4ml_amd' / 'libda4ml_hip.so').exists():
        entry.build()
    from da4ml_amd import _binary

    return _binary


def test_header_symbols_exported(hip):
    header = (ROOT / 'include' / 'da4ml_hip.h').read_text()
    declared = set(re.findall(r'\b(da_[a-z0-9_]+)\s*\(', header))
    assert declared == set(hip.EXPORTS), declared ^ set(hip.EXPORTS)
    lib = hip.lib()
    for name in declared:
        assert hasattr(lib, name), name


def test_scalar_helpers_match_oracle(hip, oracle):
    rng = np.random.default_rng(1)
    for x in list(rng.standard_normal(100).astype(np.float32)) + [0.0, 1.0, 2.0, 0.375, 1e-3, 65536.0]:
        assert hip.get_lsb_loc(float(x)) == oracle.get_lsb_loc(float(x))
        if x > 0:
            assert hip.iceil_log2(float(x)) == oracle.iceil_log2(float(x))
    for _ in range(200):
        q0 = sorted(rng.integers(-300, 300, 2).tolist())
        q1 = sorted(rng.integers(-300, 300, 2).tolist())
        q0 = (q0[0], q0[1], 2.0 ** int(rng.integers(-4, 3)))
        q1 = (q1[0], q1[1], 2.0 ** int(rng.integers(-4, 3)))
        args = (int(rng.integers(-6, 7)), bool(rng.integers(0, 2)))
        for a, c in [(-1, -1), (1, -1), (4, 8), (-1, 4)]:
            assert hip.cost_add(q0, q1, *args, a, c) == oracle.cost_add(q0, q1, *args, a, c)


def test_api_surface_matches_reference():
    """names of da4ml.cmvm / da4ml._binary (reference cmvm/__init__.py:29, _binary/__init__.py:19)"""
    import da4ml_amd.cmvm as cmvm
    from da4ml_amd import _binary

    for name in ('solve', 'QInterval', 'Op', 'CombLogic', 'kernel_decompose', 'solver_options_t'):
        assert hasattr(cmvm, name)
    for name in ('dais_interp_run', 'int_arr_to_csd', 'csd_decompose', 'get_lsb_loc', 'kernel_decompose', 'solve', 'iceil_log2'):
        assert hasattr(_binary, name)
    assert hasattr(_binary.cmvm_bin, 'cost_add')


def test_no_cpu_fallback(hip):
    if hip.device_count() > 0:
        pytest.skip('a GPU is visible; the loud-failure path is exercised on GPU-less hosts')
    k = np.eye(4, dtype=np.float32)
    with pytest.raises(RuntimeError, match='no HIP device'):
        hip.solve(k)
    with pytest.raises(RuntimeError, match='no HIP device'):
        hip.kernel_decompose(k)
    with pytest.raises(RuntimeError, match='no HIP device'):
        hip.csd_decompose(k)


def test_argument_errors(hip):
    with pytest.raises(TypeError):
        hip.solve(np.eye(3))  # float64: the reference binds kernel with .noconvert()
    with pytest.raises(TypeError):
        hip.int_arr_to_csd(np.arange(4))
    with pytest.raises(ValueError):
        hip.solve(np.eye(3, dtype=np.float32), qintervals=[(-1.0, 1.0, -0.5)] * 3)  # (0.3 is accepted: steps need not be powers of two)


def test_product_never_imports_the_oracle():
    """no file of the product package imports, links, loads or includes anything from oracle/ or tests/"""
    bad = re.compile(r'(^\s*(from|import)\s+(oracle|tests)\b)|liboracle|libref\.so|libmodel|#include\s+"[^"]*(oracle|tests/)[^"]*"|oracle/', re.M)
    for path in (ROOT / 'da4ml_amd').rglob('*'):
        if path.suffix in ('.py', '.cc', '.h', '.hip') or path.name == 'Makefile':
            hit = bad.search(path.read_text())
            assert hit is None, f'{path}: {hit.group(0)!r}'


def test_oracle_does_not_share_the_product_marshalling():
    """the checker converts C arrays to Pipeline objects with its own code (oracle/oracle.py), not with da4ml_amd._marshal"""
    src = (ROOT / 'oracle' / 'oracle.py').read_text()
    assert '_marshal' not in src.replace('da4ml_amd._marshal (row-by-row', '')


def test_error_class_comes_from_the_library_not_from_the_message():
    """da_last_error_code() tells ValueError from RuntimeError (reference: std::invalid_argument vs std::runtime_error);
    the Python shim must not sniff the message text"""
    import da4ml_amd._binary as B

    src = (ROOT / 'da4ml_amd' / '_binary' / '__init__.py').read_text()
    assert "'must' in" not in src and 'da_last_error_code' in src
    L = B.lib()
    assert L.da_last_error_code() in (0, -1, -2, -3)
    if B.device_count() == 0:  # no GPU here: set_device must fail with the NO_DEVICE class and say so
        assert L.da_set_device(0) == -3 and L.da_last_error_code() == -3


def test_sharded_entry_point_validates_its_arguments():
    """da_solve_sharded (column-sharded chains): rank / world / callback are checked before any device work -- ValueError class"""
    import ctypes as C

    import numpy as np

    import da4ml_amd._binary as B

    L = B.lib()
    k = np.eye(4, dtype=np.float32)
    st = np.zeros(3, np.int64)
    for rank, world in ((2, 2), (-1, 1), (0, 0), (0, 300)):
        h = L.da_solve_sharded(k, 4, 4, b'wmc', b'auto', -1, -2, None, None, -1, -1, 1, rank, world, None, None, st)
        assert not h and L.da_last_error_code() == -2, (rank, world)
    h = L.da_solve_sharded(k, 4, 4, b'wmc', b'auto', -1, -2, None, None, -1, -1, 1, 0, 2, None, None, st)  # world 2 needs a collective
    assert not h and L.da_last_error_code() == -2 and b'all-reduce' in L.da_last_error()
    if B.device_count() == 0:  # valid arguments, no GPU: fails loudly, no CPU path
        import pytest

        with pytest.raises(RuntimeError, match='no HIP device'):
            B.solve_sharded(k, rank=0, world=1)
