"""Result data model: replay, JSON round trip, DAIS binary header (reference src/da4ml/types.py)."""

import numpy as np

from cases import int_matrix
from da4ml_amd.types import CombLogic, Op, Pipeline, QInterval, minimal_kif


def test_replay_and_roundtrip(oracle, tmp_path):
    k = int_matrix(1, 12, 9, -32, 32)
    p = oracle.solve(k, adder_size=1, carry_size=-1)
    assert isinstance(p, Pipeline) and all(isinstance(s, CombLogic) for s in p.solutions)
    assert np.all(p.kernel == k)
    x = np.random.default_rng(0).integers(-128, 128, (50, 12)).astype(np.float64)
    assert np.array_equal(p(x), x @ k.astype(np.float64))
    assert np.array_equal(p(x[3]), x[3] @ k.astype(np.float64))
    assert np.array_equal(p.solutions[0].predict(x), p.solutions[0](x))  # DAIS binary program + integer interpreter
    from da4ml_amd._binary import dais_interp_run

    assert np.array_equal(dais_interp_run(p.solutions[0].to_binary(), x.ravel()), p.solutions[0](x))
    p.save(tmp_path / 'p.json')
    assert Pipeline.load(tmp_path / 'p.json') == p
    p.solutions[0].save(tmp_path / 's.json')
    assert CombLogic.load(tmp_path / 's.json') == p.solutions[0]
    assert p.cost == sum(s.cost for s in p.solutions)
    assert p.shape == (12, 9)
    assert len(p.out_qint) == 9 and len(p.inp_qint) == 12


def test_binary_layout(oracle):
    s = oracle.solve(int_matrix(2, 5, 4, -8, 8)).solutions[0]
    b = s.to_binary()
    assert b.dtype == np.int32
    assert list(b[:6]) == [1, 0, 5, 4, len(s.ops), 0]
    assert len(b) == 6 + 5 + 3 * 4 + 8 * len(s.ops)
    assert int(s.ref_count.sum()) >= 2 * s.n_adders


def test_minimal_kif():
    assert minimal_kif(QInterval(-128.0, 127.0, 1.0)) == (True, 7, 0)
    assert minimal_kif(QInterval(0.0, 3.0, 0.25)) == (False, 2, 2)
    assert minimal_kif(QInterval(0.0, 0.0, 1.0)) == (False, 0, 0)


def test_unsupported_opcode_raises():
    """relu & co. replay numerically (tests/test_pipeline.py); lookup tables and bitwise statements stay tracer territory"""
    s = CombLogic((1, 1), [0], [1], [0], [False], [Op(0, -1, -1, 0, QInterval(-1, 1, 1), 0.0, 0.0), Op(0, -1, 9, 0, QInterval(0, 1, 1), 0.0, 0.0)], -1, -1)
    try:
        s([1.0])
    except NotImplementedError:
        pass
    else:
        raise AssertionError('expected NotImplementedError')
    relu = s._replace(ops=[s.ops[0], Op(0, -1, 2, 0, QInterval(0, 1, 1), 0.0, 0.0)])
    assert relu([1.0]).tolist() == [1.0] and relu([-1.0]).tolist() == [0.0]
