"""Device DAIS executor (csrc/dais_gpu.hip, kernel k_dais_run) through the C ABI against the committed golden vectors of
the reference's interpreter and against the host executor.  Runs after the solver's parity tests (file name sorts last),
and every body runs in a child process: the kernel had its first GPU run after it was written (the round's GPU budget
was spent), so a device fault must fail one test, not take the pytest process -- and the solver's results -- with it."""

import gzip
import json
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from dais_cases import random_program

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _in_child(name: str):
    env_path = [str(ROOT), str(ROOT / 'tests')]
    code = f'import sys; sys.path[:0] = {env_path!r}; import test_zz_dais_gpu as t; t.{name}(); print("child ok")'
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'child ok' in r.stdout, f'{name}: rc={r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}'


def test_device_executor_golden_vectors():
    _in_child('body_golden_vectors')


def test_device_executor_equals_host_on_random_programs():
    _in_child('body_random_programs')


def test_device_executor_on_a_solver_result():
    _in_child('body_solver_result')


def body_golden_vectors():
    from da4ml_amd._binary import dais_interp_run

    gold = json.load(gzip.open(ROOT / 'tests' / 'golden' / 'dais_golden.json.gz', 'rt'))
    for case in gold['cases']:
        prog = np.asarray(case['program'], dtype=np.int32)
        x = np.asarray([float.fromhex(v) for v in case['inputs']]).reshape(case['n_samples'], -1)
        want = np.asarray([float.fromhex(v) for v in case['outputs']]).reshape(case['n_samples'], -1)
        assert np.array_equal(dais_interp_run(prog, x, executor='device'), want), case['seed']


def body_random_programs():
    from da4ml_amd._binary import dais_interp_run

    for seed in range(3000, 3060):
        prog, x = random_program(seed, n_ops=int(30 + seed % 150), n_samples=700)  # 700: a partial last block
        assert np.array_equal(dais_interp_run(prog, x, executor='device'), dais_interp_run(prog, x, n_threads=0)), seed


def body_solver_result():
    """64x64 int8 solution, 100k samples: device == host executor == the matrix product"""
    from cases import int_matrix

    from da4ml_amd.cmvm import solve

    k = int_matrix(0, 64, 64, -128, 128)
    stage = solve(k, method0='wmc', method1='wmc', decompose_dc=-1, search_all_decompose_dc=False).solutions[0]
    x = np.random.default_rng(0).integers(-128, 128, (100_000, 64)).astype(np.float64)
    y = stage.predict(x, executor='device')
    assert np.array_equal(y[:2000], stage.predict(x[:2000]))
    assert np.array_equal(y, x @ stage.kernel.astype(np.float64))
