"""Result data model: replay, JSON round trip, DAIS binary header (reference src/da4ml/types.py)."""

import numpy as np
import pytest

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


def _mux_logic():
    q = QInterval(-8.0, 7.0, 1.0)
    ops = [Op(0, -1, -1, 0, q, 0.0, 0.0), Op(1, -1, -1, 0, QInterval(0.0, 1.0, 1.0), 0.0, 0.0),
           Op(0, 0, -6, 1 | (0 << 32), QInterval(-8.0, 8.0, 1.0), 0.0, 0.0)]  # fmt: skip
    return CombLogic((2, 1), [0, 0], [2], [0], [False], ops, -1, -1)


def test_ref_count_counts_the_mux_condition():
    """reference types.py:491-493: the condition of an msb-mux statement is a reference to its statement"""
    s = _mux_logic()
    assert s.ref_count.tolist() == [2, 1, 1]


def test_debug_print_covers_every_opcode(capsys):
    s = _mux_logic()
    s([3.0, 1.0], debug=True)
    out = capsys.readouterr().out
    assert 'msb(buf[1]) ? buf[0] : -buf[0] << 0' in out and out.count('|->') == 3
    from da4ml_amd.types import _describe

    q = QInterval(0.0, 1.0, 1.0)
    for code in (-9, -6, -3, -2, -1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11):
        assert isinstance(_describe(Op(0, 0, code, 0, q, 0.0, 0.0)), str)


def test_binary_with_lookup_tables():
    """table section + pad_left encoding of to_binary (reference types.py:504-541); objects that cannot provide them raise"""

    class Table:  # the two members of the reference's LookupTable that serialisation uses
        def __init__(self, entries, pad):
            self.table, self.pad = np.asarray(entries, dtype=np.int32), pad

        def _get_pads(self, qint):
            return self.pad, 0

    q = QInterval(0.0, 3.0, 1.0)
    ops = [Op(0, -1, -1, 0, q, 0.0, 0.0), Op(0, -1, 8, 1, QInterval(-4.0, 3.0, 1.0), 0.0, 0.0)]
    s = CombLogic((1, 1), [0], [1], [0], [False], ops, -1, -1, (Table([9, 9], 0), Table([1, -2, 3, -4, 0], 1)))
    b = s.to_binary()
    assert list(b[:6]) == [1, 0, 1, 1, 2, 2]
    body = b[6 + 1 + 3 :]
    assert list(body[8 + 3 : 8 + 5]) == [1, 1]  # low word = table index, high word = pad_left
    assert list(body[16:]) == [2, 5, 9, 9, 1, -2, 3, -4, 0]
    # executed by the product's interpreter: table[1][x - min(format) - pad_left]  (x = 1..3 -> entries 0..2)
    assert s.predict(np.array([[1.0], [2.0], [3.0]])).ravel().tolist() == [1.0, -2.0, 3.0]
    for bad in (s._replace(lookup_tables=None), s._replace(lookup_tables=(object(), object()))):
        try:
            bad.to_binary()
        except NotImplementedError:
            continue
        raise AssertionError('expected NotImplementedError')


def test_predict_default_threads_from_environment(monkeypatch, oracle):
    import da4ml_amd._binary as B

    s = oracle.solve(int_matrix(2, 5, 4, -8, 8)).solutions[0]
    seen = []
    real = B.dais_interp_run
    monkeypatch.setattr(B, 'dais_interp_run', lambda prog, data, n_threads=1, executor='host': seen.append(n_threads) or real(prog, data, n_threads, executor))
    monkeypatch.setenv('DA_DEFAULT_THREADS', '3')
    x = np.zeros((4, 5))
    s.predict(x)
    s.predict(x, n_threads=2)
    assert seen == [3, 2]


def test_lazy_op_list_behaves_like_the_list_it_replaces():
    """types.OpList: the statements of a solver result are built on access; everything a caller
    can do with the reference's ``list[Op]`` gives the same answers, and the summaries do not build a single Op"""
    import json

    from da4ml_amd._marshal import stage_from_arrays
    from da4ml_amd.types import JSONEncoder, OpList

    rng = np.random.default_rng(0)
    n_in, n_out, n_ops = 3, 2, 9
    oi = np.zeros((n_ops, 4), np.int64)
    of = np.zeros((n_ops, 5), np.float32)
    for i in range(n_ops):
        oi[i] = (i, -1, -1, 0) if i < n_in else (int(rng.integers(0, i)), int(rng.integers(0, i)), int(rng.integers(0, 2)), int(rng.integers(-3, 4)))
        of[i] = (-8.0, 7.5, 0.5, float(i // 2), 0.0 if i < n_in else 1.0 + i / 4)
    args = (n_in, n_out, [0, 1, -1], [7, 8], [0, 2], [False, True], oi, of, -1, 1)
    comb = stage_from_arrays(*args)
    assert isinstance(comb.ops, OpList) and comb.ops._lazy()
    plain = [Op(*(int(v) for v in oi[i]), QInterval(*(float(v) for v in of[i, :3])), float(of[i, 3]), float(of[i, 4])) for i in range(n_ops)]
    ref = comb._replace(ops=plain)
    # summaries: from the arrays, nothing built
    assert (comb.cost, comb.n_adders, comb.latency, comb.out_latency, comb.inp_latency) == (ref.cost, ref.n_adders, ref.latency, ref.out_latency, ref.inp_latency)
    assert len(comb.ops) == n_ops and comb.ops[4] == plain[4] and comb.ops[-1] == plain[-1] and isinstance(comb.ops[4], Op) and isinstance(comb.ops[4].qint, QInterval)
    assert comb.ops._lazy()
    with pytest.raises(IndexError):
        comb.ops[n_ops]
    # equality in both directions, with lists and with another lazy view; inequality
    other = stage_from_arrays(*args)
    assert comb == other and comb.ops._lazy() and other.ops._lazy()
    assert comb == ref and ref == comb and not (comb != ref) and comb.ops == plain and plain == comb.ops
    of2 = of.copy()
    of2[5, 4] += 1
    assert stage_from_arrays(*args[:7], of2, -1, 1) != comb and stage_from_arrays(*args[:7], of2, -1, 1).ops != plain
    # iteration, slices, membership, concatenation, serialisation
    assert list(comb.ops) == plain and comb.ops[2:5] == plain[2:5] and plain[3] in comb.ops and comb.ops + [plain[0]] == plain + [plain[0]]
    assert json.dumps(comb, cls=JSONEncoder) == json.dumps(ref, cls=JSONEncoder)
    assert CombLogic.deserialize(json.loads(json.dumps(comb, cls=JSONEncoder))) == ref
    assert np.array_equal(comb.to_binary(), ref.to_binary())
    assert np.array_equal(comb(np.arange(3.0)), ref(np.arange(3.0)))
    # mutation works on the built list
    fresh = stage_from_arrays(*args)
    fresh.ops.append(plain[0])
    assert len(fresh.ops) == n_ops + 1 and fresh.ops[-1] == plain[0] and fresh.ops != plain
    fresh.ops.pop()
    assert fresh.ops == plain
    assert stage_from_arrays(n_in, 0, [0, 0, 0], [], [], [], np.zeros((0, 4), np.int64), np.zeros((0, 5), np.float32), -1, -1).ops == []
