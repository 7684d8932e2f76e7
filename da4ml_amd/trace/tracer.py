"""Dead-statement elimination on a ``CombLogic`` (reference ``src/da4ml/trace/tracer.py:161-211``)."""

from __future__ import annotations

import numpy as np

from ..types import CombLogic, Op

_NO_OPERAND = -1


def _operands(op: Op):
    """Buffer slots an op keeps alive (the msb-mux condition lives in the low word of ``data``, reference
    ``tracer.py:168``).  For an input copy ``id0`` is an *input* number, but the reference's sweep marks the buffer slot of
    that number all the same (``tracer.py:189-190``) -- in solver and tracer output the copy of input ``j`` sits in slot
    ``j``, which is how ``keep_dead_inputs`` keeps it -- so the same is done here."""
    src = [j for j in (op.id0, op.id1) if j >= 0]
    if abs(op.opcode) == 6:
        src.append(op.data & 0xFFFFFFFF)
    return src


def dead_statement_elimination(comb: CombLogic, keep_dead_inputs: bool = False) -> CombLogic:
    """Drop every op no output depends on and renumber the rest; input copies survive only if used, unless
    ``keep_dead_inputs``.  One backwards liveness sweep, as the op list is topologically ordered."""
    ops = comb.ops
    live = np.zeros(len(ops), dtype=bool)
    for idx in comb.out_idxs:
        if idx != _NO_OPERAND:
            live[idx] = True
    for i in range(len(ops) - 1, -1, -1):
        op = ops[i]
        if live[i] or (keep_dead_inputs and op.opcode == -1):
            for j in _operands(op):
                live[j] = True
    new_pos = np.cumsum(live) - 1

    def moved(op: Op) -> Op:
        if op.opcode == -1:
            return op
        id0 = int(new_pos[op.id0]) if op.id0 >= 0 else op.id0
        id1 = int(new_pos[op.id1]) if op.id1 >= 0 else op.id1
        data = op.data
        if abs(op.opcode) == 6:
            data = int(new_pos[data & 0xFFFFFFFF]) + (((data >> 32) & 0xFFFFFFFF) << 32)
        return Op(id0, id1, op.opcode, data, op.qint, op.latency, op.cost)

    kept = [moved(op) for op, alive in zip(ops, live) if alive]
    out_idxs = [int(new_pos[idx]) if idx >= 0 else -1 for idx in comb.out_idxs]
    return CombLogic(comb.shape, comb.inp_shifts, out_idxs, comb.out_shifts, comb.out_negs, kept, comb.carry_size, comb.adder_size, comb.lookup_tables)
