"""The DAIS executor behind ``dais_interp_run`` / ``CombLogic.predict`` (da4ml_amd/csrc/dais_interp.cc, C ABI ``da_dais_run``)
against the reference's own interpreter: committed golden vectors (every opcode) always, the live reference build
(oracle/_ref/libdais_ref.so) when it is present.  Host code: runs without a GPU."""

import ctypes as C
import gzip
import json
from pathlib import Path

import numpy as np
import pytest

from dais_cases import narrow_condition_program, random_program

ROOT = Path(__file__).resolve().parent.parent


def test_golden_vectors_from_the_reference():
    from da4ml_amd._binary import dais_interp_run

    gold = json.load(gzip.open(ROOT / 'tests' / 'golden' / 'dais_golden.json.gz', 'rt'))
    seen = set()
    for case in gold['cases']:
        prog = np.asarray(case['program'], dtype=np.int32)
        x = np.asarray([float.fromhex(v) for v in case['inputs']]).reshape(case['n_samples'], -1)
        want = np.asarray([float.fromhex(v) for v in case['outputs']]).reshape(case['n_samples'], -1)
        for threads in (1, 0):
            assert np.array_equal(dais_interp_run(prog, x, n_threads=threads), want), case['seed']
        # the device executor's per-thread code (dais_core.h eval + slot-compacted registers), run on the host
        assert np.array_equal(dais_interp_run(prog, x, executor='host-scalar'), want), case['seed']
        regen, _ = narrow_condition_program() if case['seed'] == 48 else random_program(case['seed'], n_samples=8)
        assert np.array_equal(regen, prog), 'tests/dais_cases.py no longer reproduces the committed programs'
        n_in, n_out, n_ops = (int(v) for v in prog[2:5])
        seen |= set(prog[6 + n_in + 3 * n_out : 6 + n_in + 3 * n_out + 8 * n_ops].reshape(-1, 8)[:, 0].tolist())
    assert seen == {-9, -6, -3, -2, -1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10}  # every opcode of docs/dais.md


def test_against_live_reference_build():
    path = ROOT / 'oracle' / '_ref' / 'libdais_ref.so'
    if not path.exists():
        pytest.skip('oracle/_ref/libdais_ref.so is not built (no /root/reference on this host)')
    from da4ml_amd._binary import dais_interp_run

    R = C.CDLL(str(path))
    R.dref_run.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    for seed in range(1000, 1300):
        prog, x = random_program(seed, n_ops=int(40 + seed % 90), n_samples=40)
        x = np.ascontiguousarray(x)
        want = np.zeros((x.shape[0], int(prog[3])))
        assert R.dref_run(prog.ctypes.data, prog.size, x.ctypes.data, x.shape[0], want.ctypes.data) == 0
        assert np.array_equal(dais_interp_run(prog, x, n_threads=2), want), seed


def test_many_samples_threaded_equals_single_thread():
    from da4ml_amd._binary import dais_interp_run

    prog, _ = random_program(7)
    x = np.random.default_rng(1).uniform(-30, 30, (5000, int(prog[2])))
    assert np.array_equal(dais_interp_run(prog, x, n_threads=0), dais_interp_run(prog, x, n_threads=1))


def test_invalid_programs_raise():
    from da4ml_amd._binary import dais_interp_run

    prog, x = random_program(3)
    bad = prog.copy()
    bad[0] = 2
    with pytest.raises(RuntimeError, match='DAIS version mismatch'):  # reference DAISInterpreter.cc:16-22
        dais_interp_run(bad, x)
    with pytest.raises(RuntimeError, match='size mismatch'):  # :43-49
        dais_interp_run(prog[:-1], x)
    n_in, n_out = int(prog[2]), int(prog[3])
    bad = prog.copy()
    first_add = next(i for i in range(int(prog[4])) if prog[6 + n_in + 3 * n_out + 8 * i] in (0, 1))
    bad[6 + n_in + 3 * n_out + 8 * first_add + 1] = first_add  # id0 = own index
    with pytest.raises(RuntimeError, match='violating causality'):  # :428-447
        dais_interp_run(bad, x)
    bad = prog.copy()
    bad[6 + n_in + 3 * n_out + 8 * first_add] = 77
    with pytest.raises(RuntimeError, match='Unknown opcode'):  # :372-376
        dais_interp_run(bad, x)


def test_solver_output_runs_through_the_executor(oracle):
    """CombLogic.predict (reference types.py:549-581) of both solved stages equals the float replay for inputs inside the
    stages' input intervals (stage 2 declares the UNSCALED intervals of stage 1's result ops, reference api.cc:100-113,
    so it is fed samples of those intervals, not stage 1's scaled outputs)"""
    from cases import int_matrix

    k = int_matrix(3, 12, 9, -64, 64)
    sol = oracle.solve(k, adder_size=1, carry_size=-1)
    rng = np.random.default_rng(0)
    for stage in sol.solutions:
        cols = []
        for q in stage.inp_qint:
            lo, hi = round(q.min / q.step), round(q.max / q.step)
            cols.append(rng.integers(lo, hi + 1, 64) * q.step)
        x = np.stack(cols, axis=1).astype(np.float64)
        assert np.array_equal(stage.predict(x), stage(x))
    x = rng.integers(-128, 128, (50, 12)).astype(np.float64)
    assert np.array_equal(sol.solutions[0].predict(x) @ sol.solutions[1].kernel.astype(np.float64), x @ k)


def test_device_code_path_on_the_host_equals_block_executor():
    """csrc/dais_core.h + the liveness slot assignment of csrc/dais_gpu.hip (what every GPU thread runs) against the
    block executor on many random programs; also a solver result, whose slot count must be far below its op count"""
    from da4ml_amd._binary import dais_interp_run

    for seed in range(2000, 2150):
        prog, x = random_program(seed, n_ops=int(30 + seed % 120), n_samples=24)
        assert np.array_equal(dais_interp_run(prog, x, executor='host-scalar'), dais_interp_run(prog, x)), seed


def test_device_executor_needs_a_gpu():
    from da4ml_amd import _binary as hip

    if hip.device_count() > 0:
        pytest.skip('a GPU is present')
    prog, x = random_program(5)
    with pytest.raises(RuntimeError, match='no HIP device'):
        hip.dais_interp_run(prog, x, executor='device')
    with pytest.raises(ValueError):
        hip.dais_interp_run(prog, x, executor='tpu')
