"""The literal statement of "drop-in": after ``da4ml_amd.install_as_da4ml()`` the reference's own test module imports
(``from da4ml._binary import csd_decompose, kernel_decompose, solve``, reference tests/test_cmvm.py:4) resolve to this
package, and the three checks of that module (tests/test_cmvm.py:23-55: digit reconstruction, m0 @ m1 == kernel for five
decompose_dc values, the 72-option solve grid with the tracer's cost model) pass on the MI355X -- same fixtures (n in
{2, 4, 8}, bits in {2, 4, 8}, uniform kernels rounded to integers), seeded instead of the reference's unseeded
``np.random.rand``."""

import itertools
import sys

import numpy as np
import pytest


@pytest.fixture(scope='module')
def da4ml_alias():
    import da4ml_amd

    before = {k: v for k, v in sys.modules.items() if k == 'da4ml' or k.startswith('da4ml.')}
    da4ml_amd.install_as_da4ml()
    yield da4ml_amd
    for k in [k for k in sys.modules if k == 'da4ml' or k.startswith('da4ml.')]:
        if k not in before:
            del sys.modules[k]


def test_alias_resolves_to_this_package(da4ml_alias):
    """CPU-checkable half: the module names the reference's callers import exist and are this package's objects"""
    import da4ml
    from da4ml._binary import csd_decompose, kernel_decompose, solve
    from da4ml.cmvm import solve as cmvm_solve
    from da4ml.trace.pipeline import to_pipeline
    from da4ml.types import CombLogic, Pipeline

    from da4ml_amd import _binary, cmvm, trace, types

    assert da4ml is da4ml_alias
    assert (csd_decompose, kernel_decompose, solve) == (_binary.csd_decompose, _binary.kernel_decompose, _binary.solve)
    assert cmvm_solve is cmvm.solve and to_pipeline is trace.to_pipeline
    assert Pipeline is types.Pipeline and CombLogic is types.CombLogic


def reference_kernel(n_dim, bits):
    rng = np.random.default_rng(977 * n_dim + bits)
    return np.round((rng.random((n_dim, n_dim)) - 0.5) * 2 ** (bits + 1)).astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize('n_dim', [2, 4, 8])
@pytest.mark.parametrize('bits', [2, 4, 8])
def test_reference_test_module_through_the_alias(da4ml_alias, n_dim, bits):
    from da4ml._binary import csd_decompose, kernel_decompose, solve

    kernel = reference_kernel(n_dim, bits)
    # test_decompose
    csd, shift0, shift1 = csd_decompose(kernel.astype(np.float32))
    shift2 = np.arange(csd.shape[-1])
    recon = csd * (2.0 ** shift0[:, None, None]) * (2.0 ** shift1[None, :, None]) * (2.0 ** shift2[None, None, :])
    assert np.all(np.sum(recon, axis=-1) == kernel)
    # test_kernel_decompose
    for dc in (-2, -1, 0, 1, 2):
        m0, m1 = kernel_decompose(kernel.astype(np.float32), dc=dc)
        assert np.all(m0 @ m1 == kernel), dc
    # test_solve: hard_dc x method0 x method1 x decompose_dc x search_all_decompose_dc, tracer cost model
    for hard_dc, method0, method1, decompose_dc, search_all in itertools.product((0, 2, -1), ('mc', 'wmc'), ('mc', 'wmc'), (0, -1, -2), (False, True)):
        sol = solve(kernel, hard_dc=hard_dc, method0=method0, method1=method1, decompose_dc=decompose_dc, search_all_decompose_dc=search_all,
                    adder_size=1, carry_size=-1)  # fmt: skip
        assert np.all(sol.kernel == kernel), (hard_dc, method0, method1, decompose_dc, search_all)
