"""Result data model of the CMVM path: ``QInterval``, ``Op``, ``CombLogic``, ``Pipeline``.

Field-for-field compatible with the reference's ``da4ml.types`` (reference
``src/da4ml/types.py:21-64`` for QInterval/Precision/Op, ``:176-215`` CombLogic fields,
``:584-633`` Pipeline) so that solver results can be swapped, JSON-dumped and diffed.
The float replay (``CombLogic.__call__``) executes what the CMVM solver emits -- opcodes -1 (input copy), 0 (add) and
1 (subtract) -- and, for numeric inputs, the arithmetic statements of the tracer (relu, quantize, constant add/definition,
msb-mux, multiply); lookup tables and bitwise operations raise there.  ``CombLogic.predict`` runs the integer-exact
DAIS executor of the native library, which implements every opcode.

Unlike the reference's per-sample Python replay (``types.py:217-370``) the numeric replay
here is vectorised over a batch, so ``CombLogic.kernel`` of a 1e5-op solution costs one
pass over the op list instead of ``n_in`` passes.
"""

from __future__ import annotations

import json
import os
from functools import reduce
from math import ceil, log2
from pathlib import Path
from collections.abc import Sequence
from typing import NamedTuple

import numpy as np

__all__ = ['QInterval', 'Precision', 'Op', 'OpList', 'Pair', 'DAState', 'CombLogic', 'Pipeline', 'minimal_kif', 'JSONEncoder']


class QInterval(NamedTuple):
    """Quantised interval ``[min, max]`` with resolution ``step``."""

    min: float
    max: float
    step: float


class Precision(NamedTuple):
    keep_negative: bool
    integers: int
    fractional: int


class Op(NamedTuple):
    """``buf[i] = buf[id0] (+|-) buf[id1] * 2**data`` for opcode 0|1; opcode -1 copies input ``id0``."""

    id0: int
    id1: int
    opcode: int
    data: int
    qint: QInterval
    latency: float
    cost: float


class OpList(Sequence):
    """The statements of a solver result as a sequence of ``Op``, built on access from the C ABI's arrays.

    The reference's binding creates one Python ``Op`` (and one ``QInterval``) per statement while the result is returned
    (reference ``_binary/cmvm/bindings.cc:106-139``): 130 k objects, 26 ms of interpreter time for a 256x256 matrix -- a batch of
    64 results costs twice what the solve itself takes on the GPU.  Here a result keeps its two arrays (``[n, 4]`` int64 =
    id0, id1, opcode, data; ``[n, 5]`` float64 = qint.min, qint.max, qint.step, latency, cost) and
    this view over them: ``len``, indexing and the summaries of ``CombLogic`` (cost, adders, latencies) read the arrays; iterating,
    slicing, comparing with a list or mutating builds the ``Op`` objects once and keeps them.  Equal to a ``list`` of the same
    ``Op`` values, serialised as one."""

    __slots__ = ('_ci', '_ff', '_ops')

    def __init__(self, ci, ff):
        self._ci, self._ff, self._ops = ci, ff, None

    @classmethod
    def from_ops(cls, ops):
        o = cls(None, None)
        o._ops = list(ops)
        return o

    # -- materialisation (the collector is paused: none of these tuples of numbers can be part of a cycle, and every allocation
    # threshold crossed would start a collection that walks everything alive -- 3 x slower, measured in round 3)
    def _all(self) -> list:
        if self._ops is None:
            import gc

            ci = self._ci.T.tolist()
            ff = self._ff
            new = tuple.__new__  # what the NamedTuple constructors end in; the field order is that of the arrays
            paused = gc.isenabled()
            gc.disable()
            try:
                qints = [new(QInterval, t) for t in map(tuple, ff[:, :3].tolist())]
                self._ops = [new(Op, t) for t in zip(ci[0], ci[1], ci[2], ci[3], qints, ff[:, 3].tolist(), ff[:, 4].tolist())] if len(ff) else []
            finally:
                if paused:
                    gc.enable()
        return self._ops

    def _lazy(self) -> bool:
        return self._ops is None

    def _mutable(self) -> list:
        ops = self._all()
        self._ci = self._ff = None  # the arrays no longer describe the list
        return ops

    # -- Sequence
    def __len__(self):
        return len(self._ci) if self._ops is None else len(self._ops)

    def __getitem__(self, i):
        if self._ops is not None or isinstance(i, slice):
            return self._all()[i]
        r, f = self._ci[i].tolist(), self._ff[i].tolist()  # (IndexError for an index out of range, negative indices as in a list)
        return Op(r[0], r[1], r[2], r[3], QInterval(f[0], f[1], f[2]), f[3], f[4])

    def __iter__(self):
        return iter(self._all())

    def __reversed__(self):
        return reversed(self._all())

    def __contains__(self, x):
        return x in self._all()

    def __eq__(self, other):
        if isinstance(other, OpList):
            if self._ops is None and other._ops is None:
                return np.array_equal(self._ci, other._ci) and np.array_equal(self._ff, other._ff)
            return self._all() == other._all()
        if isinstance(other, list):
            return self._all() == other
        return NotImplemented

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    __hash__ = None  # (a list is not hashable either)

    def __repr__(self):
        return repr(self._all())

    def to_dict(self):  # the JSON encoders write the list
        return self._all()

    def copy(self):
        return list(self._all())

    def __add__(self, other):
        return self._all() + list(other)

    def __radd__(self, other):
        return list(other) + self._all()

    # -- what a caller may do to the ``ops`` list of a reference result
    def append(self, x):
        self._mutable().append(x)

    def extend(self, xs):
        self._mutable().extend(xs)

    def insert(self, i, x):
        self._mutable().insert(i, x)

    def pop(self, i=-1):
        return self._mutable().pop(i)

    def clear(self):
        self._mutable().clear()

    def __setitem__(self, i, x):
        self._mutable()[i] = x

    def __delitem__(self, i):
        del self._mutable()[i]

    # -- summaries straight from the arrays (same arithmetic, same order as the loops over Op objects they replace)
    def _column(self, name: str) -> list:
        if self._ops is None:
            j = Op._fields.index(name)
            return self._ci[:, j].tolist() if j < 4 else self._ff[:, j - 2].tolist()  # (latency, cost = columns 3, 4 of the float array)
        return [getattr(op, name) for op in self._ops]


class Pair(NamedTuple):
    """A candidate two-term subexpression ``row[id0] +/- row[id1] << shift``."""

    id0: int
    id1: int
    sub: bool
    shift: int


class DAState(NamedTuple):
    """Snapshot of the reference's host-side solver state (reference ``types.py:76-83``); kept for import compatibility --
    the state of this implementation lives in HBM and is never materialised in this form."""

    shifts: tuple
    expr: list
    ops: list
    freq_stat: dict
    kernel: np.ndarray


def minimal_kif(qi: QInterval, symmetric: bool = False) -> Precision:
    """Smallest (sign, integer, fraction) fixed-point format holding ``qi`` (reference ``types.py:86-114``)."""
    if qi.min == qi.max == 0:
        return Precision(False, 0, 0)
    frac = int(-log2(qi.step))
    lo, hi = round(qi.min / qi.step), round(qi.max / qi.step)
    mag = max(abs(lo), hi) + 1 if symmetric else max(abs(lo), hi + 1)
    return Precision(qi.min < 0, int(ceil(log2(mag))) - frac, frac)


class _Encoder(json.JSONEncoder):
    def default(self, o):
        if hasattr(o, 'to_dict'):
            return o.to_dict()
        return super().default(o)


JSONEncoder = _Encoder  # the reference's public name (``types.py:169``)


def _is_numeric(a: np.ndarray) -> bool:
    return a.dtype != object


# Numeric replay of the statements the reference's tracer adds around solver output (reference ``types.py:250-283`` with the
# float branches of ``_relu`` / ``_quantize``, ``types.py:130-168``), vectorised over the batch: they appear in graphs that
# pass through ``da4ml_amd.trace`` (a constant definition stands for an absent output after retiming) and in loaded files.
# Lookup tables and the bitwise operations (opcodes 8-10) need the tracer's op library and stay unsupported here.
def _op_relu(ops, op, buf):
    v = buf[op.id0] if op.opcode == 2 else -buf[op.id0]
    _, i, f = minimal_kif(op.qint)
    v = np.floor(np.maximum(v, 0.0) * 2.0**f) / 2.0**f
    return v % 2.0**i


def _op_quantize(ops, op, buf):
    v = buf[op.id0] if op.opcode == 3 else -buf[op.id0]
    k, i, f = minimal_kif(op.qint)
    bits, eps = k + i + f, 2.0**-f
    bias = 2.0 ** (bits - 1) * k
    return eps * ((np.floor(v / eps) + bias) % 2.0**bits - bias)


def _op_mux(ops, op, buf):
    cond_id = op.data & 0xFFFFFFFF
    shift = (op.data >> 32) & 0xFFFFFFFF
    shift = shift if shift < 0x80000000 else shift - 0x100000000
    other = (buf[op.id1] if op.opcode == 6 else -buf[op.id1]) * 2.0**shift
    q = ops[cond_id].qint
    msb = buf[cond_id] < 0 if q.min < 0 else buf[cond_id] >= 2.0 ** (minimal_kif(q)[1] - 1)
    return np.where(msb, buf[op.id0], other)


_NUMERIC_OPS = {
    2: _op_relu,
    -2: _op_relu,
    3: _op_quantize,
    -3: _op_quantize,
    4: lambda ops, op, buf: buf[op.id0] + op.data * op.qint.step,
    5: lambda ops, op, buf: np.full(buf.shape[1], op.data * op.qint.step),
    6: _op_mux,
    -6: _op_mux,
    7: lambda ops, op, buf: buf[op.id0] * buf[op.id1],
}


def _s32(word: int) -> int:
    word &= 0xFFFFFFFF
    return word - (1 << 32) if word >= (1 << 31) else word


def _describe(op) -> str:
    """One-line description of a statement for ``CombLogic.__call__(debug=True)`` (every opcode of docs/dais.md)."""
    code, a, b, d = op.opcode, op.id0, op.id1, int(op.data)
    neg = '-' if code < 0 else ''
    if code == -1:
        return 'inp'
    if code in (0, 1):
        return f'buf[{a}] {"+-"[code]} buf[{b}]<<{d}'
    if code in (2, -2):
        return f'relu({neg}buf[{a}])'
    if code in (3, -3):
        return f'quantize({neg}buf[{a}])'
    if code == 4:
        return f'buf[{a}] + {d * op.qint.step}'
    if code == 5:
        return f'const {d * op.qint.step}'
    if code in (6, -6):
        return f'msb(buf[{d & 0xFFFFFFFF}]) ? buf[{a}] : {neg}buf[{b}] << {_s32(d >> 32)}'
    if code == 7:
        return f'buf[{a}] * buf[{b}]'
    if code == 8:
        return f'tables[{d & 0xFFFFFFFF}].lookup(buf[{a}])'
    if code in (9, -9):
        return f'{ {0: "~", 1: "any", 2: "all"}.get(d, "?")}({neg}buf[{a}])'
    if code == 10:
        sym = {0: '&', 1: '|', 2: '^'}.get((d >> 56) & 0xFF, '?')
        return f'{"-" if (d >> 32) & 1 else ""}buf[{a}] {sym} {"-" if (d >> 33) & 1 else ""}buf[{b}] << {_s32(d)}'
    return f'opcode {code}({a}, {b}, {d})'


class CombLogic(NamedTuple):
    """One combinational adder graph: ``ops`` executed in order on a buffer, then ``out_idxs`` read out."""

    shape: tuple[int, int]
    inp_shifts: list[int]
    out_idxs: list[int]
    out_shifts: list[int]
    out_negs: list[bool]
    ops: list[Op]
    carry_size: int
    adder_size: int
    lookup_tables: tuple | None = None

    # ------------------------------------------------------------------ replay
    def _replay(self, x):
        """x: [n_in] or [batch, n_in]; returns the whole buffer [n_ops] / [batch, n_ops]."""
        x = np.asarray(x)
        single = x.ndim == 1
        if single:
            x = x[None]
        numeric = _is_numeric(x)
        x = x * (2.0 ** np.asarray(self.inp_shifts, dtype=np.float64))
        n_ops = len(self.ops)
        buf = np.empty((n_ops, x.shape[0]), dtype=np.float64 if numeric else object)
        for i, op in enumerate(self.ops):
            code = op.opcode
            if code == -1:
                buf[i] = x[:, op.id0]
            elif code == 0:
                buf[i] = buf[op.id0] + buf[op.id1] * 2.0**op.data
            elif code == 1:
                buf[i] = buf[op.id0] - buf[op.id1] * 2.0**op.data
            elif numeric and code in _NUMERIC_OPS:
                buf[i] = _NUMERIC_OPS[code](self.ops, op, buf)
            else:
                raise NotImplementedError(f'opcode {code} is outside the CMVM path implemented by da4ml_amd ({op})')
        buf = buf.T
        return buf[0] if single else buf

    def __call__(self, inp, quantize=False, debug=False, dump=False):
        if quantize:  # truncate + wrap every input into the format of its input interval (reference types.py:247-249)
            inp = np.asarray(inp)
            if not _is_numeric(inp):
                raise NotImplementedError('quantize=True needs numeric inputs (symbolic replay belongs to the tracer)')
            k, i, f = (np.asarray(v, dtype=np.float64) for v in self.inp_kifs)
            bits, eps = k + i + f, 2.0**-f
            bias = 2.0 ** (bits - 1) * k
            inp = eps * ((np.floor(inp / eps) + bias) % 2.0**bits - bias)
        buf = self._replay(inp)
        if debug:
            flat = np.asarray(buf)
            flat = flat[0] if flat.ndim == 2 else flat
            for i, (op, v) in enumerate(zip(self.ops, flat)):
                print(f'{_describe(op):<48} |-> buf[{i}] = {v}')
        if dump:
            return buf
        idx = np.asarray(self.out_idxs, dtype=np.int64)
        if len(self.ops) == 0:
            return np.zeros(buf.shape[:-1] + (len(idx),))
        scale = 2.0 ** np.asarray(self.out_shifts, dtype=np.float64)
        scale = scale * np.where(np.asarray(self.out_negs, dtype=bool), -1.0, 1.0) * (idx >= 0)
        return buf[..., np.where(idx < 0, 0, idx)] * scale

    @property
    def kernel(self) -> np.ndarray:
        """The matrix this graph implements (rows = responses to one-hot inputs)."""
        n_in = self.shape[0]
        out = np.empty(self.shape, dtype=np.float32)
        chunk = max(1, min(n_in, (1 << 25) // max(len(self.ops), 1)))
        eye = np.identity(n_in)
        for lo in range(0, n_in, chunk):
            out[lo : lo + chunk] = self(eye[lo : lo + chunk])
        return out

    # ------------------------------------------------------------------ summaries
    @property
    def cost(self) -> float:
        if isinstance(self.ops, OpList):
            return float(sum(self.ops._column('cost')))  # (the same left-to-right sum)
        return float(sum(op.cost for op in self.ops))

    @property
    def n_adders(self) -> int:
        """Number of two-input adders/subtractors (opcode 0 or 1)."""
        if isinstance(self.ops, OpList):
            return sum(1 for c in self.ops._column('opcode') if c in (0, 1))
        return sum(1 for op in self.ops if op.opcode in (0, 1))

    @property
    def latency(self) -> tuple[float, float]:
        lat = [self.ops[i].latency for i in self.out_idxs]
        return (min(lat), max(lat)) if lat else (0.0, 0.0)

    @property
    def out_latency(self) -> list[float]:
        return [self.ops[i].latency if i >= 0 else 0.0 for i in self.out_idxs]

    @property
    def out_qint(self) -> list[QInterval]:
        res = []
        for i, idx in enumerate(self.out_idxs):
            lo, hi, st = self.ops[idx].qint
            sf = 2.0 ** self.out_shifts[i]
            lo, hi, st = lo * sf, hi * sf, st * sf
            if self.out_negs[i]:
                lo, hi = -hi, -lo
            res.append(QInterval(lo, hi, st))
        return res

    @property
    def out_kifs(self):
        return np.array([minimal_kif(q) for q in self.out_qint]).T

    @property
    def inp_latency(self) -> list[float]:
        if isinstance(self.ops, OpList):
            return [lat for lat, c in zip(self.ops._column('latency'), self.ops._column('opcode')) if c == -1]
        return [op.latency for op in self.ops if op.opcode == -1]

    @property
    def inp_qint(self) -> list[QInterval]:
        q = [QInterval(0.0, 0.0, 1.0)] * self.shape[0]
        for op in self.ops:
            if op.opcode == -1:
                q[op.id0] = op.qint
        return q

    @property
    def inp_kifs(self):
        return np.array([minimal_kif(q) for q in self.inp_qint]).T

    @property
    def ref_count(self) -> np.ndarray:
        cnt = np.zeros(len(self.ops), dtype=np.uint64)
        for op in self.ops:
            if op.opcode == -1:
                continue
            for j in (op.id0, op.id1):
                if j != -1:
                    cnt[j] += 1
            if op.opcode in (6, -6):  # the msb-mux condition is a third operand (reference types.py:491-493)
                cnt[op.data & 0xFFFFFFFF] += 1
        for i in self.out_idxs:
            if i >= 0:
                cnt[i] += 1
        return cnt

    def __repr__(self):
        lo, hi = self.latency
        return f'Solution([{self.shape[0]} -> {self.shape[1]}], cost={self.cost}, latency={lo}-{hi})'

    # ------------------------------------------------------------------ serialisation (reference types.py:442-477, 500-541)
    def save(self, path: str | Path):
        with open(path, 'w') as f:
            json.dump(self, f, cls=_Encoder, separators=(',', ':'))

    @classmethod
    def deserialize(cls, data: list):
        assert len(data) in (8, 9), len(data)
        ops = [Op(*o[:4], QInterval(*o[4]), *o[5:]) for o in data[5]]
        if len(data) > 8 and data[8] is not None:
            raise NotImplementedError('lookup tables are outside the CMVM path')
        return cls(tuple(data[0]), data[1], data[2], data[3], data[4], ops, data[6], data[7], None)

    @classmethod
    def load(cls, path: str | Path):
        with open(path) as f:
            return cls.deserialize(json.load(f))

    def to_binary(self, version: int = 0) -> np.ndarray:
        """int32 DAIS program (layout: reference ``docs/dais.md:70-95``, ``types.py:500-541``): header, 8 words per
        statement, then -- when the logic carries lookup tables -- the table sizes and the tables.  A table object must
        offer what the reference's ``LookupTable`` does (``.table`` int32 entries, ``._get_pads(qint)``); anything else
        raises instead of writing a program whose table statements could not be executed."""
        n_in, n_out = self.shape
        tables = self.lookup_tables
        if tables is not None and not all(hasattr(t, 'table') and hasattr(t, '_get_pads') for t in tables):
            raise NotImplementedError('lookup tables without .table / ._get_pads cannot be serialised to a DAIS program')
        head = np.concatenate(
            [[1, version, n_in, n_out, len(self.ops), 0 if tables is None else len(tables)], self.inp_shifts, self.out_idxs, self.out_shifts, self.out_negs]
        ).astype(np.int32)
        code = np.zeros((len(self.ops), 8), dtype=np.int32)
        for i, op in enumerate(self.ops):
            code[i, 0:3] = (op.opcode, op.id0, op.id1)
            data = int(op.data)
            if op.opcode == 8:  # table statement: index in the low word, left padding of the table in the high word
                if tables is None:
                    raise NotImplementedError(f'statement {i} looks up table {data} but the logic has no lookup_tables')
                data = (int(tables[data]._get_pads(self.ops[op.id0].qint)[0]) << 32) | data
            code[i, 3:5].view(np.uint64)[0] = np.uint64(data & 0xFFFFFFFFFFFFFFFF)
            code[i, 5:8] = minimal_kif(op.qint)
        words = [head, code.ravel()]
        if tables is not None:
            bodies = [np.asarray(t.table, dtype=np.int32).ravel() for t in tables]
            words += [np.asarray([len(b) for b in bodies], dtype=np.int32), *bodies]
        return np.concatenate(words)

    def save_binary(self, path: str | Path, version: int = 0):
        self.to_binary(version).tofile(str(path))

    def predict(self, data, n_threads: int = 0, executor: str = 'host'):
        """Integer-exact batch execution through the DAIS binary program, like the reference (``types.py:549-581``);
        ``executor='device'`` runs the samples on the GPU (one thread per sample) instead of on host threads."""
        from ._binary import dais_interp_run

        if isinstance(data, (list, tuple)):
            data = np.concatenate([np.asarray(a).reshape(len(a), -1) for a in data], axis=-1)
        if n_threads <= 0:  # the reference's default thread count comes from the environment (types.py:577-578)
            n_threads = int(os.environ.get('DA_DEFAULT_THREADS', 0))
        return dais_interp_run(self.to_binary(), np.asarray(data, dtype=np.float64), n_threads, executor)


class Pipeline(NamedTuple):
    """Cascade of CombLogic stages; ``solve`` returns the two stages ``x @ m0`` then ``@ m1``."""

    solutions: tuple[CombLogic, ...]

    def __call__(self, inp, quantize=False, debug=False):
        out = np.asarray(inp)
        for sol in self.solutions:
            out = sol(out, quantize=quantize, debug=debug)
        return out

    @property
    def kernel(self):
        return reduce(lambda a, b: a @ b, [s.kernel for s in self.solutions])

    @property
    def cost(self):
        return sum(s.cost for s in self.solutions)

    @property
    def n_adders(self) -> int:
        return sum(s.n_adders for s in self.solutions)

    @property
    def latency(self):
        return self.solutions[-1].latency

    @property
    def inp_qint(self):
        return self.solutions[0].inp_qint

    @property
    def inp_latency(self):
        return self.solutions[0].inp_latency

    @property
    def out_qint(self):
        return self.solutions[-1].out_qint

    @property
    def out_latencies(self):
        return self.solutions[-1].out_latency

    @property
    def shape(self):
        return self.solutions[0].shape[0], self.solutions[-1].shape[1]

    @property
    def inp_shifts(self):
        return self.solutions[0].inp_shifts

    @property
    def out_shift(self):
        return self.solutions[-1].out_shifts

    @property
    def out_neg(self):
        return self.solutions[-1].out_negs

    def __repr__(self):
        dims = ' -> '.join(str(s.shape[0]) for s in self.solutions) + f' -> {self.shape[1]}'
        lo, hi = self.latency
        return f'CascatedSolution([{dims}], cost={self.cost}, latency={lo}-{hi})'

    def save(self, path: str | Path):
        with open(path, 'w') as f:
            json.dump(self, f, cls=_Encoder, separators=(',', ':'))

    @classmethod
    def deserialize(cls, data):
        return cls(tuple(CombLogic.deserialize(s) for s in data[0]))

    @classmethod
    def load(cls, path: str | Path):
        with open(path) as f:
            return cls.deserialize(json.load(f))

    @property
    def reg_bits(self):
        bits = sum(sum(minimal_kif(q)) for q in self.inp_qint)
        for s in self.solutions:
            bits += sum(sum(minimal_kif(q)) for q in s.out_qint)
        return bits
