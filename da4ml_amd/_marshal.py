"""SoA result buffers -> ``da4ml_amd.types`` objects.

The C-ABI returns each stage as flat arrays (``ops_i`` [n_ops,4] int64 = id0,id1,opcode,data and
``ops_f`` [n_ops,5] float32 = qint.min,qint.max,qint.step,latency,cost).  The reference's binding builds
Python ``Op`` NamedTuples one by one in C++ while it returns (reference ``bindings.cc:106-139``); here the stage keeps the two
arrays behind a sequence view (``types.OpList``) that builds the objects on access -- SURVEY.md section 8f rank 2: a caller
that only wants cost, adder count and latencies of a 65 k-statement stage no longer pays 26 ms of interpreter time for
130 k objects it never looks at.
"""

from __future__ import annotations

import numpy as np

from .types import CombLogic, OpList, Pipeline


def stage_from_arrays(n_in, n_out, inp_shifts, out_idxs, out_shifts, out_negs, ops_i, ops_f, carry_size, adder_size):
    ci = np.ascontiguousarray(np.asarray(ops_i, dtype=np.int64).reshape(-1, 4))
    ff = np.asarray(ops_f, dtype=np.float32).reshape(-1, 5).astype(np.float64)  # fp32 values widened, as the reference returns them
    return CombLogic(
        (int(n_in), int(n_out)),
        np.asarray(inp_shifts, dtype=np.int64).tolist(),
        np.asarray(out_idxs, dtype=np.int64).tolist(),
        np.asarray(out_shifts, dtype=np.int64).tolist(),
        [bool(v) for v in np.asarray(out_negs).tolist()],
        OpList(ci, ff),  # Op objects are built when somebody looks at them (types.OpList)
        int(carry_size),
        int(adder_size),
    )


def pipeline_from_stages(stages):
    return Pipeline(tuple(stages))
