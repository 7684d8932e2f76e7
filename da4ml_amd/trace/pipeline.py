"""Register-stage splitting of solver output: ``to_pipeline`` and ``retime_pipeline``.

Same names, arguments and results as the reference's ``src/da4ml/trace/pipeline.py`` (``to_pipeline`` :63-167,
``retime_pipeline`` :8-31).  The reference retimes by replaying the pipeline on its symbolic tracer
(``FixedVariable``, 1.2 kLoC) and tracing the result back; the tracer is out of scope here, so ``_Retimer`` below
re-derives exactly what that round trip produces for the statements the CMVM solver emits -- input copies, add and
subtract: intervals and power-of-two scale factors of every value (``fixed_variable.py:441-513,586-609``), the adder
cost/latency model with the stage-boundary rule (``fixed_variable.py:341-379``), the tracer's statement order
(``tracer.py:12-58``) and its dead-statement pass, with constant terms (zero-width input intervals become constant-add
statements, absent outputs the constant zero).  Any other opcode raises ``NotImplementedError`` instead of guessing.  Pinned to the reference's own Python (run in the build container, see
``make_pipeline_golden.py`` next to the golden vectors) by ``pipeline_golden.json.gz``.
"""

from __future__ import annotations

from math import ceil, floor, log2

from ..types import CombLogic, Op, Pipeline, QInterval
from .tracer import dead_statement_elimination

_OUT_MARK = -1001  # second operand of the pseudo statements that stand for the outputs while splitting


def _stage_of(latency: float, cutoff: float) -> int:
    return floor(latency / (cutoff + 1e-9)) if cutoff > 0 else 0


class _Splitter:
    """Distributes the statements of one CombLogic over register stages.

    ``slots[i]`` maps stage -> position of value ``i`` in that stage's op list.  A value used in a later stage than the
    last one it is available in is carried forward: it becomes an output of every stage in between and an input copy
    (latency = the stage boundary, cost 0) of the following one."""

    def __init__(self, values: list[Op], cutoff: float):
        self.values = values
        self.cutoff = cutoff
        self.stage_ops: dict[int, list[Op]] = {}
        self.stage_outs: dict[int, list[int]] = {}
        self.slots: list[dict[int, int]] = []

    def place(self, stage: int, op: Op) -> int:
        lst = self.stage_ops.setdefault(stage, [])
        lst.append(op)
        return len(lst) - 1

    def emit(self, stage: int, pos: int) -> int:
        lst = self.stage_outs.setdefault(stage, [])
        lst.append(pos)
        return len(lst) - 1

    def operand(self, i: int, stage: int) -> int:
        if i < 0:
            return i
        have = self.slots[i]
        if stage in have:
            return have[stage]
        last = max(have)
        pos = have[last]
        for s in range(last, stage):
            port = self.emit(s, have[s])
            pos = self.place(s + 1, Op(port, -1, -1, 0, self.values[i].qint, float(self.cutoff * (s + 1)), 0.0))
            have[s + 1] = pos
        return pos

    def run(self):
        for i, op in enumerate(self.values):
            stage = _stage_of(op.latency, self.cutoff)
            if op.opcode == -1:
                self.slots.append({stage: self.place(stage, op)})
                continue
            a = self.operand(op.id0, stage)
            b = self.operand(op.id1, stage)
            data = op.data
            if op.opcode in (6, -6):  # msb-mux: the condition's slot is the low word of data
                cond = self.operand(data & 0xFFFFFFFF, stage)
                data = ((data >> 32) & 0xFFFFFFFF) << 32 | cond
            if b == _OUT_MARK:
                self.emit(stage, a)
            else:
                self.slots.append({stage: self.place(stage, Op(a, b, op.opcode, data, op.qint, op.latency, op.cost))})


def _split(comb: CombLogic, cutoff: float) -> Pipeline:
    assert len(comb.ops) > 0, 'No operations in the record'
    values = list(comb.ops)
    # all outputs leave in the stage of the slowest one (an absent output, index -1, reads the last statement here, as in
    # the reference where ops[-1] is Python's last element)
    t_out = max(values[i].latency for i in comb.out_idxs)
    for i in comb.out_idxs:
        values.append(Op(i, _OUT_MARK, _OUT_MARK, 0, comb.ops[i].qint, t_out, 0.0))
    sp = _Splitter(values, cutoff)
    sp.run()

    stages = []
    last = max(sp.stage_ops)
    n_in = comb.shape[0]
    for s in range(len(sp.stage_ops)):
        ops, outs = sp.stage_ops[s], sp.stage_outs[s]
        final = s == last
        tables = None
        if comb.lookup_tables is not None:  # every stage keeps only the tables it uses, renumbered in ascending order
            used = sorted({op.data for op in ops if op.opcode == 8})
            renumber = {t: j for j, t in enumerate(used)}
            ops = [op._replace(data=renumber[op.data]) if op.opcode == 8 else op for op in ops]
            tables = tuple(comb.lookup_tables[t] for t in used)
        stages.append(
            CombLogic(
                shape=(n_in, len(outs)),
                inp_shifts=[0] * n_in,
                out_idxs=outs,
                out_shifts=comb.out_shifts if final else [0] * len(outs),
                out_negs=comb.out_negs if final else [False] * len(outs),
                ops=ops,
                carry_size=comb.carry_size,
                adder_size=comb.adder_size,
                lookup_tables=tables,
            )
        )
        n_in = len(outs)
    return Pipeline(tuple(stages))


def to_pipeline(comb: CombLogic, latency_cutoff: float, retiming: bool = True, verbose: bool = True) -> Pipeline:
    """Split ``comb`` into register stages: a statement goes to stage ``floor(latency / latency_cutoff)``.

    ``retiming`` (default, like the reference) afterwards searches the smallest cutoff that still gives the same number
    of stages, re-deriving all latencies under that cutoff (``retime_pipeline``); ``verbose`` prints the cutoff found."""
    csol = _split(comb, latency_cutoff)
    if retiming:
        csol = retime_pipeline(csol, verbose=verbose)
    return csol


# ----------------------------------------------------------------------------------------------------------- retiming


def _const_frac_bits(value: float) -> int:
    """Smallest f in (-32, 32] with ``value * 2**f`` integral; -32 for zero (reference ``fixed_variable.py:201-216``)."""
    value = float(value)
    if value == 0:
        return -32
    for f in range(-31, 32):
        scaled = value * 2.0**f
        if scaled == int(scaled):
            return f
    return 32


class _RetimeInfeasible(Exception):
    """An adder alone is slower than the cutoff (the reference's tracer asserts there, ``fixed_variable.py:376``)."""


class _Retimer:
    """Re-derives a pipeline of add/subtract statements under a new latency cutoff.

    A *term* is ``(node, low, high, step, factor)``: a node of the new graph seen through a power-of-two (possibly
    negative) scale, with the interval of the scaled value.  Nodes: ``src[n]`` is ``None`` for an input or a constant,
    the one term a constant was added to (``cadd[n]``), else the two terms that were added.  All numbers are dyadic rationals well inside double precision, so floats are exact where the
    reference uses ``Decimal``."""

    def __init__(self, adder_size: int, carry_size: int, cutoff: float, cost_add):
        self.adder_size, self.carry_size, self.cutoff = adder_size, carry_size, cutoff
        self.cost_add = cost_add
        self.src: list[tuple | None] = []
        self.lat: list[float] = []
        self.cost: list[float] = []
        self.const: list[float | None] = []  # value of a constant node (zero-width interval), None otherwise
        self.cadd: dict[int, float] = {}  # constant-add nodes: node -> constant in units of the source term (src[n] = (term,))

    # -- terms
    def new_input(self, qint: QInterval):
        lo, hi, st = (float(v) for v in qint)
        assert lo <= hi, f'low {lo} must be less than high {hi}'
        self.src.append(None)
        self.lat.append(0.0)
        self.cost.append(0.0)
        self.const.append(lo if lo == hi else None)
        return (len(self.src) - 1, lo, hi, st, 1.0)

    def new_const(self, value: float):
        """A fresh constant (``FixedVariable.from_const``): its own node, latency and cost 0, factor 1."""
        self.src.append(None)
        self.lat.append(0.0)
        self.cost.append(0.0)
        self.const.append(float(value))
        return (len(self.src) - 1, value, value, 2.0 ** -_const_frac_bits(value), 1.0)

    def scaled(self, t, c: float):
        n, lo, hi, st, f = t
        if self.const[n] is not None:
            return self.new_const(self.const[n] * c)
        if c == 0:
            return self.new_const(0.0)
        if log2(abs(c)) % 1 != 0:
            raise NotImplementedError(f'scale {c} is not a power of two')
        a, b = lo * c, hi * c
        return (n, min(a, b), max(a, b), abs(st * c), f * c)

    def negated(self, t):
        n, lo, hi, st, f = t
        if self.const[n] is not None:
            return self.new_const(-self.const[n])
        return (n, -hi, -lo, st, -f)

    def const_added(self, t, c: float):
        """``t + c`` for a plain number ``c`` (reference ``_const_add``, fixed_variable.py:484-513): nothing for zero, a new
        constant for a constant, a constant-add statement otherwise -- merged into ``t``'s own when ``t`` already is one."""
        n, lo, hi, st, f = t
        if c == 0:
            return t
        if self.const[n] is not None:
            return self.new_const(self.const[n] + c)
        if n in self.cadd:
            parent = self.src[n][0]
            sf = f / parent[4]
            return self.scaled(self.const_added(parent, self.cadd[n] * parent[4] + c / sf), sf)
        data = c / f
        fb = _const_frac_bits(data)
        self.src.append((t,))
        self.lat.append(self.lat[n])
        self.cost.append(float(ceil(log2(abs(data) + 2.0**-fb))) + fb)
        self.const.append(None)
        self.cadd[len(self.src) - 1] = data
        return (len(self.src) - 1, lo + c, hi + c, min(st, 2.0 ** -_const_frac_bits(c)), f)

    def added(self, a, b):
        if self.const[b[0]] is not None:
            return self.const_added(a, self.const[b[0]])
        if self.const[a[0]] is not None:
            return self.const_added(b, self.const[a[0]])
        if a[4] < 0:
            if b[4] > 0:
                return self.added(b, a)
            return self.negated(self.added(self.negated(a), self.negated(b)))
        lo, hi, st = a[1] + b[1], a[2] + b[2], min(a[3], b[3])
        assert lo < hi, 'sum of two non-constant terms cannot be constant'
        dlat, cost = self.cost_add((a[1], a[2], a[3]), (b[1], b[2], b[3]), 0, False, self.adder_size, self.carry_size)
        base = max(self.lat[a[0]], self.lat[b[0]])
        lat = dlat + base
        cut = self.cutoff
        if cut > 0 and ceil(lat / cut) > ceil(base / cut):
            if not dlat <= cut:
                raise _RetimeInfeasible
            lat = ceil(base / cut) * cut + dlat
        self.src.append((a, b))
        self.lat.append(lat)
        self.cost.append(cost)
        self.const.append(None)
        return (len(self.src) - 1, lo, hi, st, a[4])

    # -- replay of the given pipeline (reference types.py:217-372 executed on FixedVariables)
    def replay(self, comb: CombLogic, inp: list):
        inp = [self.scaled(t, 2.0**s) for t, s in zip(inp, comb.inp_shifts)]
        buf: list = [None] * len(comb.ops)
        for i, op in enumerate(comb.ops):
            if op.opcode == -1:
                buf[i] = inp[op.id0]
            elif op.opcode in (0, 1):
                rhs = self.scaled(buf[op.id1], 2.0**op.data)
                buf[i] = self.added(buf[op.id0], rhs if op.opcode == 0 else self.negated(rhs))
            else:
                raise NotImplementedError(f'retiming covers what the CMVM solver emits (opcodes -1, 0, 1), not opcode {op.opcode}')
        out = []
        for idx, sh, neg in zip(comb.out_idxs, comb.out_shifts, comb.out_negs):
            # an absent output (index -1) reads the last statement and is masked to the constant zero, like the reference
            t = self.scaled(buf[idx], 2.0**sh)
            out.append(self.scaled(self.scaled(t, -1.0 if neg else 1.0), 0.0 if idx < 0 else 1.0))
        return out

    # -- back to a statement list (reference tracer.py:12-58, 61-158, 214-250)
    def trace(self, inputs: list, outputs: list) -> CombLogic:
        order: list[int] = []  # nodes in the tracer's gathering order: inputs, then post-order from each output
        seen = set()
        for t in inputs:
            seen.add(t[0])
            order.append(t[0])
        for t in outputs:
            stack = [(t[0], False)]
            while stack:
                n, expanded = stack.pop()
                if expanded:
                    order.append(n)
                    continue
                if n in seen:
                    continue
                seen.add(n)
                stack.append((n, True))
                if self.src[n] is None:  # a constant output
                    continue
                for t in reversed(self.src[n]):
                    stack.append((t[0], False))
        total = len(order)
        ranked = sorted(range(total), key=lambda i: self.lat[order[i]] * total + i)
        order = [order[i] for i in ranked]
        # statements nothing refers to disappear before numbering, inputs stay (constant inputs cannot be referred to here)
        used = {t[0] for t in outputs}
        for n in order:
            if self.src[n] is not None:
                used.update(t[0] for t in self.src[n])
        input_no = {t[0]: j for j, t in enumerate(inputs)}
        order = [n for n in order if n in used or n in input_no]
        index = {n: i for i, n in enumerate(order)}
        shape_of = {t[0]: t for t in inputs}
        ops: list[Op] = []
        for i, n in enumerate(order):
            if self.const[n] is not None:  # constant definition (a constant *input* is never referred to: dead, removed below)
                v = self.const[n]
                step = 2.0 ** -_const_frac_bits(v)
                ops.append(Op(-1, -1, 5, int(v / step), QInterval(v, v, step), 0.0, 0.0))
                continue
            if self.src[n] is None:
                _, lo, hi, st, f = shape_of[n]
                ops.append(Op(input_no[n], -1, -1, 0, QInterval(lo, hi, st), self.lat[n], 0.0))
                continue
            if n in self.cadd:  # constant add: data in units of the statement's own (unscaled) step
                (a,) = self.src[n]
                c = self.cadd[n] * a[4]
                lo, hi = sorted(((a[1] + c) / a[4], (a[2] + c) / a[4]))
                st = min(a[3], 2.0 ** -_const_frac_bits(c)) / abs(a[4])
                assert index[a[0]] < i
                ops.append(Op(index[a[0]], -1, 4, int(self.cadd[n] / st), QInterval(lo, hi, st), self.lat[n], self.cost[n]))
                continue
            a, b = self.src[n]
            fa, fb = a[4], b[4]
            lo, hi, st = (a[1] + b[1]) / fa, (a[2] + b[2]) / fa, min(a[3], b[3]) / fa  # fa > 0: the unscaled interval
            assert index[a[0]] < i and index[b[0]] < i
            ops.append(Op(index[a[0]], index[b[0]], int(fb < 0), int(log2(abs(fb / fa))), QInterval(lo, hi, st), self.lat[n], self.cost[n]))
        facs = [t[4] for t in outputs]
        comb = CombLogic((len(inputs), len(outputs)), [0] * len(inputs), [index[t[0]] for t in outputs], [int(log2(abs(f))) for f in facs],
                         [f < 0 for f in facs], ops, self.carry_size, self.adder_size, None)  # fmt: skip
        return dead_statement_elimination(comb)


def _retimed(csol: Pipeline, cutoff: float, cost_add) -> Pipeline | None:
    first = csol.solutions[0]
    rt = _Retimer(first.adder_size, first.carry_size, cutoff, cost_add)
    inputs = [rt.new_input(q) for q in csol.inp_qint]
    terms = inputs
    try:
        for stage in csol.solutions:
            terms = rt.replay(stage, terms)
    except _RetimeInfeasible:
        return None
    return _split(rt.trace(inputs, terms), cutoff)


def retime_pipeline(csol: Pipeline, verbose: bool = True) -> Pipeline:
    """Bisect the latency cutoff down to the smallest (to within 1) that keeps the number of stages, re-deriving every
    latency under each trial cutoff (an adder that would cross a stage boundary starts at the boundary instead).
    One deliberate deviation: where the reference's bisection stops making progress and loops forever (integral lower
    bound, upper bound less than 2 above it, midpoint infeasible) this returns the best pipeline found so far."""
    from .._binary import cost_add

    n_stages = len(csol.solutions)
    hi = max(max(sol.out_latency) / (i + 1) for i, sol in enumerate(csol.solutions))
    lo = max(csol.out_latencies) / n_stages
    best = csol
    while hi - lo > 1:
        cutoff = (hi + lo) // 2
        trial = _retimed(csol, cutoff, cost_add)
        if trial is None or len(trial.solutions) > n_stages:
            if cutoff == lo:
                # (hi + lo) // 2 == lo (integral lo, 1 < hi - lo < 2) and still too fast: the state cannot change any
                # more.  The reference's loop (pipeline.py:15-28) never terminates here; stop with the best found
                break
            lo = cutoff
        else:
            hi, best = cutoff, trial
    if verbose:
        print(f'actual cutoff: {hi}')
    return best
