"""SoA result buffers -> ``da4ml_amd.types`` objects.

The C-ABI returns each stage as flat arrays (``ops_i`` [n_ops,4] int64 = id0,id1,opcode,data and
``ops_f`` [n_ops,5] float32 = qint.min,qint.max,qint.step,latency,cost).  The reference's binding builds
Python ``Op`` NamedTuples one by one in C++ (reference ``bindings.cc:106-139``); here the conversion is
one ``tolist()`` per array COLUMN and two comprehensions that create the tuples directly, with the cyclic garbage
collector paused meanwhile: a 65 k-op stage is 130 k new container objects, each allocation threshold crossed starts a
collection that walks everything alive (the earlier results included), and none of these tuples of numbers can be part of a
cycle -- 3 x faster than the same loop with the collector running (32 vs 75-120 ms per 65 k-op stage here).
"""

from __future__ import annotations

import gc

import numpy as np

from .types import CombLogic, Op, Pipeline, QInterval


def stage_from_arrays(n_in, n_out, inp_shifts, out_idxs, out_shifts, out_negs, ops_i, ops_f, carry_size, adder_size):
    ci = np.asarray(ops_i, dtype=np.int64).reshape(-1, 4).T.tolist()
    ff = np.asarray(ops_f, dtype=np.float32).reshape(-1, 5).astype(np.float64)  # fp32 values widened, as the reference returns them
    new = tuple.__new__  # what the NamedTuple constructors end in; the field order is that of the arrays
    paused = gc.isenabled()
    gc.disable()
    try:
        qints = [new(QInterval, t) for t in map(tuple, ff[:, :3].tolist())]
        ops = [new(Op, t) for t in zip(ci[0], ci[1], ci[2], ci[3], qints, ff[:, 3].tolist(), ff[:, 4].tolist())]
    finally:
        if paused:
            gc.enable()
    return CombLogic(
        (int(n_in), int(n_out)),
        np.asarray(inp_shifts, dtype=np.int64).tolist(),
        np.asarray(out_idxs, dtype=np.int64).tolist(),
        np.asarray(out_shifts, dtype=np.int64).tolist(),
        [bool(v) for v in np.asarray(out_negs).tolist()],
        ops,
        int(carry_size),
        int(adder_size),
    )


def pipeline_from_stages(stages):
    return Pipeline(tuple(stages))
