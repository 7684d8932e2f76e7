"""ctypes front-end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this module;
the product package ``da4ml_amd`` never does.

Two back-ends share one C ABI shape:
  * ``Oracle('port')``  -> oracle/liboracle.so   (plain-C++ restatement, prefix ``orc_``)
  * ``Oracle('ref')``   -> oracle/_ref/libref.so (the real reference sources built against the shim, prefix ``ref_``)
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

from da4ml_amd.types import CombLogic, Op, Pipeline, QInterval  # the result data model only: the conversion below is the oracle's own

HERE = Path(__file__).resolve().parent
_f32p = np.ctypeslib.ndpointer(np.float32, flags='C_CONTIGUOUS')
_i64p = np.ctypeslib.ndpointer(np.int64, flags='C_CONTIGUOUS')
_i8p = np.ctypeslib.ndpointer(np.int8, flags='C_CONTIGUOUS')
_i32p = np.ctypeslib.ndpointer(np.int32, flags='C_CONTIGUOUS')



def stage_from_arrays(n_in, n_out, inp_shifts, out_idxs, out_shifts, out_negs, ops_i, ops_f, carry_size, adder_size):
    """C arrays of one stage -> CombLogic.  Deliberately NOT the product's da4ml_amd._marshal (row-by-row here, array-wise
    there): a conversion bug must not cancel between the checker and the thing checked."""
    ops = []
    for row in range(len(ops_i)):
        id0, id1, opcode, data = (int(ops_i[row, c]) for c in range(4))
        lo, hi, step, latency, cost = (float(ops_f[row, c]) for c in range(5))
        ops.append(Op(id0, id1, opcode, data, QInterval(lo, hi, step), latency, cost))
    as_ints = lambda a: [int(v) for v in a]  # noqa: E731
    return CombLogic((int(n_in), int(n_out)), as_ints(inp_shifts), as_ints(out_idxs), as_ints(out_shifts), [int(v) != 0 for v in out_negs],
                     ops, int(carry_size), int(adder_size))  # fmt: skip


def pipeline_from_stages(stages):
    return Pipeline(tuple(stages))


STAT_NAMES = ('iterations', 'p_init', 'd0', 'f_first', 'f_sum', 'f_max', 'live_sum', 'match_sum', 'regen_pairs', 'tree_ops')


def build(targets=('liboracle.so', 'ref')):
    subprocess.run(['make', '-C', str(HERE), *targets], check=True, stdout=subprocess.DEVNULL)


class Oracle:
    def __init__(self, kind: str = 'port'):
        self.kind = kind
        path, self.p = {
            'port': (HERE / 'liboracle.so', 'orc_'),
            'ref': (HERE / '_ref' / 'libref.so', 'ref_'),
            'model': (HERE.parent / 'tests' / 'model' / 'libmodel.so', 'mdl_'),  # sequential model of the GPU engine
        }[kind]
        if kind == 'model':
            subprocess.run(['make', '-C', str(path.parent), 'libmodel.so'], check=True, stdout=subprocess.DEVNULL)
        elif not path.exists():
            build(('liboracle.so',) if kind == 'port' else ('ref',))
        if not path.exists():
            raise FileNotFoundError(f'{path} is not built (kind={kind})')
        self.lib = L = C.CDLL(str(path))
        g = lambda n: getattr(L, self.p + n)  # noqa: E731
        g('last_error').restype = C.c_char_p
        g('get_lsb_loc').argtypes = [C.c_float]
        g('iceil_log2').argtypes = [C.c_float]
        g('cost_add').argtypes = [_f32p, _f32p, C.c_int64, C.c_int, C.c_int, C.c_int, _f32p]
        g('int_arr_to_csd').argtypes = [_i32p, C.c_int64, C.c_void_p]
        g('csd_decompose').argtypes = [_f32p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        g('kernel_decompose').argtypes = [_f32p, C.c_int64, C.c_int64, C.c_int, _f32p, _f32p]
        g('solve').restype = C.c_void_p
        g('solve').argtypes = [_f32p, C.c_int64, C.c_int64, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]  # fmt: skip
        g('n_stages').argtypes = [C.c_void_p]
        g('picked').argtypes = [C.c_void_p]
        g('stage_info').argtypes = [C.c_void_p, C.c_int, _i64p]
        g('stage_copy').argtypes = [C.c_void_p, C.c_int, _i64p, _i64p, _i64p, _i64p, _i64p, _f32p]
        g('free').argtypes = [C.c_void_p]
        if kind in ('port', 'ref'):
            g('sample_chain').argtypes = [_f32p, C.c_int64, C.c_int64, C.c_char_p, C.c_double, np.ctypeslib.ndpointer(np.float64)]
        if kind == 'port':
            g('solve_single').restype = C.c_void_p
            g('solve_single').argtypes = [_f32p, C.c_int64, C.c_int64, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
            g('stage_stats').argtypes = [C.c_void_p, C.c_int, _i64p]

        self.g = g

    # ---- scalar helpers -------------------------------------------------
    def get_lsb_loc(self, x: float) -> int:
        return int(self.g('get_lsb_loc')(x))

    def iceil_log2(self, x: float) -> int:
        return int(self.g('iceil_log2')(x))

    def cost_add(self, q0, q1, shift: int, sub: bool, adder_size: int, carry_size: int):
        out = np.zeros(2, np.float32)
        self.g('cost_add')(np.asarray(q0, np.float32), np.asarray(q1, np.float32), shift, int(sub), adder_size, carry_size, out)
        return float(out[0]), float(out[1])

    # ---- decompositions -------------------------------------------------
    def int_arr_to_csd(self, x):
        x = np.ascontiguousarray(x, dtype=np.int32)
        N = self.g('int_arr_to_csd')(x.ravel(), x.size, None)
        out = np.zeros(x.shape + (N,), np.int8)
        self.g('int_arr_to_csd')(x.ravel(), x.size, out.ctypes.data)
        return out

    def csd_decompose(self, kernel, center: bool = True):
        k = np.ascontiguousarray(kernel, dtype=np.float32)
        n_in, n_out = k.shape
        N = self.g('csd_decompose')(k, n_in, n_out, int(center), None, None, None)
        csd = np.zeros((n_in, n_out, N), np.int8)
        s0, s1 = np.zeros(n_in, np.int8), np.zeros(n_out, np.int8)
        self.g('csd_decompose')(k, n_in, n_out, int(center), csd.ctypes.data, s0.ctypes.data, s1.ctypes.data)
        return csd, s0, s1

    def kernel_decompose(self, kernel, dc: int = -2):
        k = np.ascontiguousarray(kernel, dtype=np.float32)
        n_in, n_out = k.shape
        m0, m1 = np.zeros((n_in, n_out), np.float32), np.zeros((n_out, n_out), np.float32)
        self.g('kernel_decompose')(k, n_in, n_out, dc, m0, m1)
        return m0, m1

    # ---- solve ----------------------------------------------------------
    def _collect(self, h, stats=False):
        if not h:
            raise RuntimeError(self.g('last_error')().decode())
        try:
            stages, st = [], []
            for s in range(self.g('n_stages')(h)):
                info = np.zeros(5, np.int64)
                self.g('stage_info')(h, s, info)
                n_in, n_out, n_ops, carry, adder = (int(v) for v in info)
                a = [np.zeros(n, np.int64) for n in (n_in, n_out, n_out, n_out)]
                oi, of = np.zeros((n_ops, 4), np.int64), np.zeros((n_ops, 5), np.float32)
                self.g('stage_copy')(h, s, *a, oi, of)
                stages.append(stage_from_arrays(n_in, n_out, *a, oi, of, carry, adder))
                if stats and self.kind == 'port':
                    sv = np.zeros(10, np.int64)
                    self.g('stage_stats')(h, s, sv)
                    st.append(dict(zip(STAT_NAMES, sv.tolist())))
            picked = int(self.g('picked')(h))
        finally:
            self.g('free')(h)
        pipe = pipeline_from_stages(stages)
        return (pipe, st, picked) if stats else pipe

    @staticmethod
    def _opt(qintervals, latencies, n_in):
        q = None if qintervals is None else np.ascontiguousarray(np.asarray(qintervals, np.float32).reshape(n_in, 3))
        l = None if latencies is None else np.ascontiguousarray(np.asarray(latencies, np.float32).reshape(n_in))
        return q, l

    def solve(self, kernel, method0='wmc', method1='auto', hard_dc=-1, decompose_dc=-2, qintervals=None, latencies=None,
              adder_size=-1, carry_size=-1, search_all_decompose_dc=True, stats=False):  # fmt: skip
        k = np.ascontiguousarray(kernel, dtype=np.float32)
        n_in, n_out = k.shape
        q, l = self._opt(qintervals, latencies, n_in)
        h = self.g('solve')(k, n_in, n_out, method0.encode(), method1.encode(), hard_dc, decompose_dc,
                            None if q is None else q.ctypes.data, None if l is None else l.ctypes.data,
                            adder_size, carry_size, int(search_all_decompose_dc))  # fmt: skip
        return self._collect(h, stats)

    def solve_many(self, kernels, method0='wmc', method1='auto', hard_dc=-1, decompose_dc=-2, qintervals=None, latencies=None,
                   adder_size=-1, carry_size=-1, search_all_decompose_dc=True):  # fmt: skip
        """Batch through the product's host logic (``da::solve_batch``) on the sequential engine model; 'model' only.
        ``qintervals`` / ``latencies``: None or one entry (possibly None) per kernel."""
        assert self.kind == 'model'
        ks = [np.ascontiguousarray(k, dtype=np.float32) for k in kernels]
        n = len(ks)
        qs, ls = [], []
        for i, k in enumerate(ks):
            q, l = self._opt(None if qintervals is None else qintervals[i], None if latencies is None else latencies[i], k.shape[0])
            qs.append(q)
            ls.append(l)
        ptr = lambda arrs: (C.c_void_p * n)(*[None if a is None else a.ctypes.data for a in arrs])  # noqa: E731
        fn = self.lib.mdl_solve_batch
        fn.argtypes = [C.c_int, C.c_void_p, _i64p, _i64p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]  # fmt: skip
        res = (C.c_void_p * n)()
        rc = fn(n, ptr(ks), np.array([k.shape[0] for k in ks], np.int64), np.array([k.shape[1] for k in ks], np.int64), method0.encode(),
                method1.encode(), hard_dc, decompose_dc, ptr(qs), ptr(ls), adder_size, carry_size, int(search_all_decompose_dc), res)  # fmt: skip
        if rc != 0:
            raise RuntimeError(self.g('last_error')().decode())
        return [self._collect(res[i]) for i in range(n)]

    def solve_sharded(self, kernel, method0='wmc', method1='auto', hard_dc=-1, decompose_dc=-2, qintervals=None, latencies=None,
                      adder_size=-1, carry_size=-1, search_all_decompose_dc=True, rank=0, world=1, allreduce=None):  # fmt: skip
        """Column-sharded solve through the PRODUCT's orchestration (csrc/cmvm_shard.cc) on the sequential engine model;
        'model' only.  ``allreduce``: ctypes callback from da4ml_amd.multi_gpu.make_allreduce_callback.  Returns
        (Pipeline, {sharded_chains, greedy_steps, allreduce_calls})."""
        assert self.kind == 'model'
        k = np.ascontiguousarray(kernel, dtype=np.float32)
        n_in, n_out = k.shape
        q, l = self._opt(qintervals, latencies, n_in)
        fn = self.lib.mdl_solve_sharded
        fn.restype = C.c_void_p
        fn.argtypes = [_f32p, C.c_int64, C.c_int64, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                       C.c_int, C.c_int, C.c_void_p, C.c_void_p, _i64p]  # fmt: skip
        st = np.zeros(3, np.int64)
        h = fn(k, n_in, n_out, method0.encode(), method1.encode(), hard_dc, decompose_dc, None if q is None else q.ctypes.data,
               None if l is None else l.ctypes.data, adder_size, carry_size, int(search_all_decompose_dc), rank, world,
               C.cast(allreduce, C.c_void_p), None, st)  # fmt: skip
        if not h:
            raise RuntimeError('column-sharded solve of the engine model failed')
        return self._collect(h), dict(zip(('sharded_chains', 'greedy_steps', 'allreduce_calls'), st.tolist()))

    def comm_abort(self):
        """what an all-reduce callback's owner calls when a collective failed (the model's da_comm_abort)"""
        self.lib.mdl_comm_abort()

    def chains_run(self, reset=True) -> int:
        """Chains handed to the model backend since the last reset ('model' only)."""
        self.lib.mdl_chains_run.restype = C.c_longlong
        return int(self.lib.mdl_chains_run(int(reset)))

    def solve_single(self, kernel, method='wmc', qintervals=None, latencies=None, adder_size=-1, carry_size=-1, stats=False):
        """One greedy chain + adder tree on ``kernel`` (reference cmvm_core.cc:227-237); 'port' only."""
        k = np.ascontiguousarray(kernel, dtype=np.float32)
        n_in, n_out = k.shape
        q, l = self._opt(qintervals, latencies, n_in)
        h = self.g('solve_single')(k, n_in, n_out, method.encode(), None if q is None else q.ctypes.data,
                                   None if l is None else l.ctypes.data, adder_size, carry_size)  # fmt: skip
        r = self._collect(h, stats)
        return (r[0].solutions[0], r[1][0]) if stats else r.solutions[0]


def sample_chain(oracle: Oracle, kernel, method: str, budget_s: float) -> dict:
    """time-bounded CPU sample of one greedy chain (state creation + as many iterations as fit in ``budget_s``)"""
    k = np.ascontiguousarray(kernel, dtype=np.float32)
    out = np.zeros(4, np.float64)
    if oracle.g('sample_chain')(k, k.shape[0], k.shape[1], method.encode(), float(budget_s), out) != 0:
        raise RuntimeError(oracle.g('last_error')().decode())
    return dict(create_s=float(out[0]), iterations=int(out[1]), iter_s=float(out[2]), finished=bool(out[3]))


def set_threads(n: int):
    os.environ['OMP_NUM_THREADS'] = str(n)
