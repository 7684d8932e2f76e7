"""SoA result buffers -> ``da4ml_amd.types`` objects.

The C-ABI returns each stage as flat arrays (``ops_i`` [n_ops,4] int64 = id0,id1,opcode,data and
``ops_f`` [n_ops,5] float32 = qint.min,qint.max,qint.step,latency,cost).  The reference's binding builds
Python ``Op`` NamedTuples one by one in C++ (reference ``bindings.cc:106-139``); here the conversion is
one ``tolist()`` per array plus a single comprehension, keeping the Python-side cost linear and small.
"""

from __future__ import annotations

import numpy as np

from .types import CombLogic, Op, Pipeline, QInterval


def stage_from_arrays(n_in, n_out, inp_shifts, out_idxs, out_shifts, out_negs, ops_i, ops_f, carry_size, adder_size):
    ii = np.asarray(ops_i, dtype=np.int64).reshape(-1, 4).tolist()
    ff = np.asarray(ops_f, dtype=np.float32).reshape(-1, 5).astype(np.float64).tolist()
    ops = [Op(a[0], a[1], a[2], a[3], QInterval(b[0], b[1], b[2]), b[3], b[4]) for a, b in zip(ii, ff)]
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
