"""ctypes binding of ``libda4ml_hip.so`` -- the counterpart of the reference's ``da4ml._binary``
(src/da4ml/_binary/__init__.py:4-19), which imports the nanobind module ``cmvm_bin``.

Same names, argument meaning, defaults and exception classes as the reference's bindings
(src/da4ml/_binary/cmvm/bindings.cc:227-264).  There is deliberately no CPU implementation behind these
functions: if the HIP library or a GPU is missing they raise.
"""

from __future__ import annotations

import ctypes as C
from pathlib import Path
from types import SimpleNamespace

import numpy as np

from .._marshal import pipeline_from_stages, stage_from_arrays

import os as _os

_LIB_PATH = Path(_os.environ['DA4ML_HIP_LIB']).resolve() if _os.environ.get('DA4ML_HIP_LIB') else Path(__file__).resolve().parent.parent / 'libda4ml_hip.so'
_lib = None

_f32p = np.ctypeslib.ndpointer(np.float32, flags='C_CONTIGUOUS')
_i64p = np.ctypeslib.ndpointer(np.int64, flags='C_CONTIGUOUS')
_i32p = np.ctypeslib.ndpointer(np.int32, flags='C_CONTIGUOUS')

EXPORTS = (
    'da_last_error da_last_error_code da_version da_device_count da_set_device da_get_lsb_loc da_iceil_log2 da_cost_add da_int_arr_to_csd '
    'da_csd_decompose da_kernel_decompose da_solve da_solve_batch da_solve_sharded da_rccl_unique_id da_rccl_shutdown da_solve_sharded_rccl da_comm_abort da_shard_exchanged_elements da_n_stages da_picked da_stage_info da_stage_copy '
    'da_result_stats da_free da_timings da_engine_stats da_dais_run da_dais_last_error da_dais_run_on'
).split()


def lib():
    """Load (once) and return the C-ABI library; raises ImportError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # the greedy loop's launches are queued by two host threads that never block (csrc/cmvm_engine.hip, HipBackend::run_chains): with
    # fewer than three cores to run on they would take turns on one another's time slices -- one thread then (the backend reads this when
    # it is created; an explicit setting wins)
    try:
        if len(_os.sched_getaffinity(0)) < 3:
            _os.environ.setdefault('DA4ML_HIP_LAUNCH_THREADS', '1')
    except (AttributeError, OSError):
        pass
    csrc = Path(__file__).resolve().parent.parent / 'csrc'
    sources = [p for p in csrc.glob('*') if p.suffix in ('.hip', '.cc', '.h') or p.name == 'Makefile'] + [_LIB_PATH.parent.parent / 'include' / 'da4ml_hip.h']
    def _stale():
        return not _LIB_PATH.exists() or any(p.exists() and p.stat().st_mtime > _LIB_PATH.stat().st_mtime for p in sources)

    if not _os.environ.get('DA4ML_HIP_LIB') and _stale():
        # build in-tree (hipcc cross-compiles for gfx950 without a GPU); never fall back to anything else.  Ranks of one
        # job reach this point together: the build is serialised by a file lock, goes to a temporary file and is moved
        # into place atomically, so that no process ever maps a half-written library; a failed rebuild is an error, not
        # a silent load of the stale library (its ABI may no longer match the argtypes below).
        import fcntl
        import shutil
        import subprocess

        if shutil.which('make') and (shutil.which('hipcc') or Path('/opt/rocm/bin/hipcc').exists()):
            with open(_LIB_PATH.parent / '.build.lock', 'w') as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                if _stale():  # not rebuilt by another process while this one waited
                    tmp = _LIB_PATH.with_suffix(f'.tmp{_os.getpid()}.so')
                    r = subprocess.run(['make', '-C', str(csrc), 'variant', f'OUT={tmp}'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                    if r.returncode != 0 or not tmp.exists():
                        tmp.unlink(missing_ok=True)
                        raise ImportError(f'rebuilding {_LIB_PATH.name} failed (sources are newer than the library):\n{r.stdout[-2000:]}')
                    _os.replace(tmp, _LIB_PATH)
    if not _LIB_PATH.exists():
        raise ImportError(f'{_LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` or `make -C da4ml_amd/csrc`')
    L = C.CDLL(str(_LIB_PATH))
    L.da_last_error.restype = C.c_char_p
    L.da_last_error_code.restype = C.c_int
    L.da_version.restype = C.c_char_p
    L.da_set_device.argtypes = [C.c_int]
    L.da_get_lsb_loc.argtypes = [C.c_float]
    L.da_iceil_log2.argtypes = [C.c_float]
    L.da_cost_add.argtypes = [_f32p, _f32p, C.c_int64, C.c_int, C.c_int, C.c_int, _f32p]
    L.da_int_arr_to_csd.argtypes = [_i32p, C.c_int64, C.c_void_p]
    L.da_csd_decompose.argtypes = [_f32p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.da_kernel_decompose.argtypes = [_f32p, C.c_int64, C.c_int64, C.c_int, _f32p, _f32p]
    L.da_solve.restype = C.c_void_p
    L.da_solve.argtypes = [_f32p, C.c_int64, C.c_int64, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.da_solve_batch.argtypes = [C.c_int, C.c_void_p, _i64p, _i64p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]  # fmt: skip
    L.da_solve_sharded.restype = C.c_void_p
    L.da_solve_sharded.argtypes = [_f32p, C.c_int64, C.c_int64, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_void_p, C.c_void_p, _i64p]  # fmt: skip
    L.da_n_stages.argtypes = [C.c_void_p]
    L.da_picked.argtypes = [C.c_void_p]
    L.da_stage_info.argtypes = [C.c_void_p, C.c_int, _i64p]
    L.da_stage_copy.argtypes = [C.c_void_p, C.c_int, _i64p, _i64p, _i64p, _i64p, _i64p, _f32p]
    L.da_result_stats.argtypes = [C.c_void_p, _i64p]
    L.da_free.argtypes = [C.c_void_p]
    L.da_rccl_unique_id.argtypes = [C.c_char_p]
    L.da_rccl_shutdown.restype = C.c_int
    L.da_rccl_shutdown.argtypes = []
    if hasattr(L, 'da_shard_exchanged_elements'):  # (libraries of earlier revisions loaded through DA4ML_HIP_LIB for A/B timing lack it; __graft_entry__.build() and tests/test_abi.py check the product's exports)
        L.da_shard_exchanged_elements.restype = C.c_int64
        L.da_shard_exchanged_elements.argtypes = []
    L.da_solve_sharded_rccl.restype = C.c_void_p
    L.da_solve_sharded_rccl.argtypes = [_f32p, C.c_int64, C.c_int64, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_char_p, _i64p]
    L.da_timings.argtypes = [np.ctypeslib.ndpointer(np.float64, flags='C_CONTIGUOUS'), C.c_int]
    L.da_engine_stats.argtypes = [np.ctypeslib.ndpointer(np.float64, flags='C_CONTIGUOUS'), C.c_int]
    _lib = L
    return L


def _raise(rc: int):
    msg = lib().da_last_error().decode()
    if rc == -2:
        raise ValueError(msg)
    raise RuntimeError(msg)


def _kernel(kernel) -> np.ndarray:
    # the reference binds `kernel` as nb::ndarray<float>.noconvert(): anything but float32 is a TypeError
    if not isinstance(kernel, np.ndarray) or kernel.dtype != np.float32:
        raise TypeError('kernel must be a numpy.ndarray of dtype float32 (no implicit conversion)')
    return np.ascontiguousarray(kernel)


def device_count() -> int:
    return int(lib().da_device_count())


def set_device(device: int) -> None:
    if lib().da_set_device(int(device)) != 0:
        _raise(-1)


def get_lsb_loc(x: float) -> int:
    return int(lib().da_get_lsb_loc(float(x)))


def iceil_log2(x: float) -> int:
    return int(lib().da_iceil_log2(float(x)))


def cost_add(q0, q1, shift: int, sub: bool, adder_size: int, carry_size: int) -> tuple[float, float]:
    out = np.zeros(2, np.float32)
    lib().da_cost_add(np.asarray(q0, np.float32), np.asarray(q1, np.float32), int(shift), int(bool(sub)), int(adder_size), int(carry_size), out)
    return float(out[0]), float(out[1])


def int_arr_to_csd(inp) -> np.ndarray:
    if not isinstance(inp, np.ndarray) or inp.dtype != np.int32:
        raise TypeError('inp must be a numpy.ndarray of dtype int32 (no implicit conversion)')
    x = np.ascontiguousarray(inp)
    N = lib().da_int_arr_to_csd(x.ravel(), x.size, None)
    if N < 0:
        _raise(N)
    out = np.zeros(x.shape + (N,), np.int8)
    rc = lib().da_int_arr_to_csd(x.ravel(), x.size, out.ctypes.data)
    if rc < 0:
        _raise(rc)
    return out


def csd_decompose(inp, center: bool = True):
    k = _kernel(inp)
    if k.ndim != 2:
        raise RuntimeError('csd_decompose only supports 2D arrays.')
    n_in, n_out = k.shape
    N = lib().da_csd_decompose(k, n_in, n_out, int(center), None, None, None)
    if N < 0:
        _raise(N)
    csd = np.zeros((n_in, n_out, N), np.int8)
    s0, s1 = np.zeros(n_in, np.int8), np.zeros(n_out, np.int8)
    rc = lib().da_csd_decompose(k, n_in, n_out, int(center), csd.ctypes.data, s0.ctypes.data, s1.ctypes.data)
    if rc < 0:
        _raise(rc)
    return csd, s0, s1


def kernel_decompose(kernel, dc: int = -2):
    k = _kernel(kernel)
    if k.ndim != 2:
        raise RuntimeError('csd_decompose only supports 2D arrays.')
    n_in, n_out = k.shape
    m0, m1 = np.zeros((n_in, n_out), np.float32), np.zeros((n_out, n_out), np.float32)
    rc = lib().da_kernel_decompose(k, n_in, n_out, int(dc), m0, m1)
    if rc < 0:
        _raise(rc)
    return m0, m1


def _collect(h, with_stats=False):
    L = lib()
    try:
        stages = []
        for s in range(L.da_n_stages(h)):
            info = np.zeros(5, np.int64)
            L.da_stage_info(h, s, info)
            n_in, n_out, n_ops, carry, adder = (int(v) for v in info)
            arr = [np.zeros(n, np.int64) for n in (n_in, n_out, n_out, n_out)]
            oi, of = np.zeros((n_ops, 4), np.int64), np.zeros((n_ops, 5), np.float32)
            L.da_stage_copy(h, s, *arr, oi, of)
            stages.append(stage_from_arrays(n_in, n_out, *arr, oi, of, carry, adder))
        pipe = pipeline_from_stages(stages)
        if not with_stats:
            return pipe
        st = np.zeros(8, np.int64)
        L.da_result_stats(h, st)
        names = ('iterations', 'digits0', 'blocks0', 'select_rounds', 'table_peak', 'scan_slots', 'partners', 'matches')
        return pipe, dict(zip(names, st.tolist()), picked=int(L.da_picked(h)))
    finally:
        L.da_free(h)


def _opt_arrays(qintervals, latencies, n_in):
    q = l = None
    if qintervals is not None:
        q = np.ascontiguousarray(np.asarray([tuple(float(v) for v in t) for t in qintervals], np.float32).reshape(n_in, 3))
    if latencies is not None:
        l = np.ascontiguousarray(np.asarray([float(v) for v in latencies], np.float32).reshape(n_in))
    return q, l


def solve(
    kernel,
    method0: str = 'wmc',
    method1: str = 'auto',
    hard_dc: int = -1,
    decompose_dc: int = -2,
    qintervals=None,
    latencies=None,
    adder_size: int = -1,
    carry_size: int = -1,
    search_all_decompose_dc: bool = True,
    _stats: bool = False,
):
    """Optimise ``x @ kernel`` into a two-stage shift-add adder graph (reference ``cmvm_bin.solve``)."""
    k = _kernel(kernel)
    if k.ndim != 2:
        raise RuntimeError('csd_decompose only supports 2D arrays.')
    n_in, n_out = k.shape
    q, l = _opt_arrays(qintervals, latencies, n_in)
    h = lib().da_solve(k, n_in, n_out, method0.encode(), method1.encode(), int(hard_dc), int(decompose_dc),
                       None if q is None else q.ctypes.data, None if l is None else l.ctypes.data,
                       int(adder_size), int(carry_size), int(bool(search_all_decompose_dc)))  # fmt: skip
    if not h:
        _raise(lib().da_last_error_code())
    return _collect(h, _stats)


def solve_sharded(kernel, method0: str = 'wmc', method1: str = 'auto', hard_dc: int = -1, decompose_dc: int = -2, qintervals=None, latencies=None,
                  adder_size: int = -1, carry_size: int = -1, search_all_decompose_dc: bool = True, rank: int = 0, world: int = 1, allreduce=None):  # fmt: skip
    """``solve`` with every greedy chain sharded over the output columns of its matrix across ``world`` processes
    (``da_solve_sharded``, include/da4ml_hip.h).  ``allreduce``: ctypes callback ``void(ctx, buf, count, on_device)``, see
    ``da4ml_amd.multi_gpu.solve_column_sharded`` (the user-facing entry).  Returns (Pipeline, exchange statistics)."""
    k = _kernel(kernel)
    if k.ndim != 2:
        raise RuntimeError('csd_decompose only supports 2D arrays.')
    n_in, n_out = k.shape
    q, l = _opt_arrays(qintervals, latencies, n_in)
    st = np.zeros(3, np.int64)
    h = lib().da_solve_sharded(k, n_in, n_out, method0.encode(), method1.encode(), int(hard_dc), int(decompose_dc),
                               None if q is None else q.ctypes.data, None if l is None else l.ctypes.data, int(adder_size), int(carry_size),
                               int(bool(search_all_decompose_dc)), int(rank), int(world), C.cast(allreduce, C.c_void_p) if allreduce is not None else None, None, st)  # fmt: skip
    if not h:
        _raise(lib().da_last_error_code())
    return _collect(h), dict(zip(('sharded_chains', 'greedy_steps', 'allreduce_calls'), st.tolist()))


def rccl_unique_id() -> bytes:
    """128-byte RCCL unique id (``ncclGetUniqueId``): rank 0 obtains it and hands it to every rank of a column-sharded solve."""
    buf = C.create_string_buffer(128)
    rc = lib().da_rccl_unique_id(buf)
    if rc != 0:
        _raise(rc)
    return buf.raw


def rccl_shutdown() -> int:
    """Destroy the RCCL communicators the library keeps (one per unique id it was handed) and free their staging buffers;
    returns how many there were.  Every rank calls it, with no solve running and the process group still alive."""
    n = lib().da_rccl_shutdown()
    if n < 0:
        _raise(1)
    return n


def solve_sharded_rccl(kernel, unique_id: bytes, method0: str = 'wmc', method1: str = 'auto', hard_dc: int = -1, decompose_dc: int = -2, qintervals=None,
                       latencies=None, adder_size: int = -1, carry_size: int = -1, search_all_decompose_dc: bool = True, rank: int = 0, world: int = 1):  # fmt: skip
    """``solve_sharded`` over the library's own RCCL transport (``da_solve_sharded_rccl``): ``ncclAllReduce`` in place on the
    library's HIP stream, no Python in the loop.  ``unique_id``: the 128 bytes rank 0 got from ``rccl_unique_id()``."""
    k = _kernel(kernel)
    if k.ndim != 2:
        raise RuntimeError('csd_decompose only supports 2D arrays.')
    if len(unique_id) != 128:
        raise ValueError('unique_id must be the 128 bytes of rccl_unique_id()')
    n_in, n_out = k.shape
    q, l = _opt_arrays(qintervals, latencies, n_in)
    st = np.zeros(3, np.int64)
    h = lib().da_solve_sharded_rccl(k, n_in, n_out, method0.encode(), method1.encode(), int(hard_dc), int(decompose_dc),
                                    None if q is None else q.ctypes.data, None if l is None else l.ctypes.data, int(adder_size), int(carry_size),
                                    int(bool(search_all_decompose_dc)), int(rank), int(world), unique_id, st)  # fmt: skip
    if not h:
        _raise(lib().da_last_error_code())
    return _collect(h), dict(zip(('sharded_chains', 'greedy_steps', 'allreduce_calls'), st.tolist()))


def shard_exchanged_elements() -> int:
    """int32 elements the last column-sharded solve of this process handed to all-reduce(sum) (x 4 = bytes per rank and direction)"""
    return int(lib().da_shard_exchanged_elements())


def comm_abort():
    """To be called by the owner of the all-reduce callback when a collective failed: the running ``solve_sharded`` stops at
    its next exchange with a RuntimeError instead of going on with a buffer that was not reduced."""
    lib().da_comm_abort()


def solve_many(
    kernels,
    method0: str = 'wmc',
    method1: str = 'auto',
    hard_dc: int = -1,
    decompose_dc: int = -2,
    qintervals=None,
    latencies=None,
    adder_size: int = -1,
    carry_size: int = -1,
    search_all_decompose_dc: bool = True,
    _stats: bool = False,
):
    """Solve independent matrices concurrently on the GPU (addition to the reference API; same options as ``solve``).

    ``qintervals`` / ``latencies`` are ``None`` or per-kernel lists (entries may be ``None``).
    """
    ks = [_kernel(k) for k in kernels]
    n = len(ks)
    if n == 0:
        return []
    n_in = np.array([k.shape[0] for k in ks], np.int64)
    n_out = np.array([k.shape[1] for k in ks], np.int64)
    kptr = (C.c_void_p * n)(*[k.ctypes.data for k in ks])
    keep, qptr, lptr = [], None, None
    if qintervals is not None or latencies is not None:
        qs, ls = [], []
        for i in range(n):
            q, l = _opt_arrays(None if qintervals is None else qintervals[i], None if latencies is None else latencies[i], int(n_in[i]))
            keep += [q, l]
            qs.append(None if q is None else q.ctypes.data)
            ls.append(None if l is None else l.ctypes.data)
        qptr, lptr = (C.c_void_p * n)(*qs), (C.c_void_p * n)(*ls)
    res = (C.c_void_p * n)()
    rc = lib().da_solve_batch(n, kptr, n_in, n_out, method0.encode(), method1.encode(), int(hard_dc), int(decompose_dc), qptr, lptr,
                              int(adder_size), int(carry_size), int(bool(search_all_decompose_dc)), res)  # fmt: skip
    if rc != 0:
        _raise(rc)
    return [_collect(res[i], _stats) for i in range(n)]


class RawBatch:
    """Handles of a solved batch kept on the C side (no Python objects built): the benchmark times the C-ABI call
    itself and converts / inspects results outside the timed region."""

    def __init__(self, handles):
        self.handles = list(handles)

    def __len__(self):
        return len(self.handles)

    def summary(self, i: int) -> dict:
        """(cost, adders, ops per stage) of result ``i`` straight from the C arrays"""
        L, h = lib(), self.handles[i]
        cost, adders, n_ops = 0.0, 0, []
        for s in range(L.da_n_stages(h)):
            info = np.zeros(5, np.int64)
            L.da_stage_info(h, s, info)
            n_in, n_out, n = int(info[0]), int(info[1]), int(info[2])
            arr = [np.zeros(k, np.int64) for k in (n_in, n_out, n_out, n_out)]
            oi, of = np.zeros((n, 4), np.int64), np.zeros((n, 5), np.float32)
            L.da_stage_copy(h, s, *arr, oi, of)
            cost += float(of[:, 4].astype(np.float64).sum())
            adders += int(np.isin(oi[:, 2], (0, 1)).sum())
            n_ops.append(n)
        st = np.zeros(8, np.int64)
        L.da_result_stats(h, st)
        return dict(cost=cost, adders=adders, n_ops=n_ops, iterations=int(st[0]))

    def pipeline(self, i: int):
        h, self.handles[i] = self.handles[i], None
        return _collect(h)

    def free(self):
        for h in self.handles:
            if h:
                lib().da_free(h)
        self.handles = []


def solve_many_raw(kernels, method0='wmc', method1='auto', hard_dc=-1, decompose_dc=-2, adder_size=-1, carry_size=-1,
                   search_all_decompose_dc=True) -> RawBatch:  # fmt: skip
    """``solve_many`` without result marshalling (benchmark use)."""
    ks = [_kernel(k) for k in kernels]
    n = len(ks)
    n_in = np.array([k.shape[0] for k in ks], np.int64)
    n_out = np.array([k.shape[1] for k in ks], np.int64)
    kptr = (C.c_void_p * n)(*[k.ctypes.data for k in ks])
    res = (C.c_void_p * n)()
    rc = lib().da_solve_batch(n, kptr, n_in, n_out, method0.encode(), method1.encode(), int(hard_dc), int(decompose_dc), None, None,
                              int(adder_size), int(carry_size), int(bool(search_all_decompose_dc)), res)  # fmt: skip
    if rc != 0:
        _raise(rc)
    return RawBatch([res[i] for i in range(n)])


def timings(reset: bool = False) -> dict:
    """Accumulated device-side timings / counters of the greedy loops (benchmark instrumentation).  ``found`` / ``inserts`` (blocks the update
    kernel found / created) are tallied only by calls made under ``DA4ML_HIP_STATS=1`` (read at every call): the tally costs 1.8 % of a step."""
    t = np.zeros(32, np.float64)
    e = np.zeros(16, np.float64)
    lib().da_engine_stats(e, 16)  # before the (possibly resetting) da_timings
    rc = lib().da_timings(t, int(reset))
    if rc != 0:
        _raise(rc)
    names = ('loop_ms', 'dist_ms', 'total_ms', 'lockstep_iters', 'iterations', 'rescans', 'partners', 'chains', 'table_bytes', 'arena_bytes',
             'select_ms_sampled', 'update_ms_sampled', 'samples', 'found', 'inserts', 'cell_reads', 'block_bytes', 'cell_bytes',
             'sel_load_bounds', 'sel_argmax', 'sel_newrow', 'sel_substitute', 'sel_prefix', 'sel_claims', 'sel_special',
             'upd_fetch', 'upd_probe', 'upd_cells', 'upd_blocks', 'upd_create', 'retries', 'sampled_chain_launches')
    extra = ('select_bytes', 'host_launch_ms', 'fast_steps', 'search_bounds', 'search_argmax', 'search_excluded', 'search_steps_timed', 'search_stale_rereads', 'search_touch_rereads', 'search_rounds', 'search_long_lists', 'search_longest_list', 'search_full_passes', 'search_list_entries')
    return {**dict(zip(names, t.tolist())), **dict(zip(extra, e.tolist()))}


_DAIS_EXECUTORS = {'host': 0, 'cpu': 0, 'device': 1, 'gpu': 1, 'host-scalar': 2}


def dais_interp_run(bin_logic, data, n_threads: int = 1, executor: str = 'host'):
    """Integer-exact execution of a DAIS program (``CombLogic.to_binary()``; layout: reference ``docs/dais.md:70-95``) on a
    batch of samples: ``data`` float64 with ``n_samples * n_in`` elements -> float64 [n_samples, n_out].  Replaces
    ``dais_bin.run_interp`` (reference ``_binary/dais/bindings.cc:102-131``); the executor is ``da_dais_run`` in
    ``libda4ml_hip.so`` (``csrc/dais_interp.cc``, all opcodes, host threads over the samples).  Host utility for checking
    solutions -- the reference's interpreter runs on the host too; not part of the solver path.

    ``executor`` (no reference counterpart): ``'host'`` (default), ``'device'`` = the HIP kernel ``k_dais_run`` (one
    thread per sample, ``csrc/dais_gpu.hip``; raises without a GPU), ``'host-scalar'`` = the device executor's per-thread
    code run on the host (test aid).  All three give identical results."""
    prog = np.ascontiguousarray(np.ravel(bin_logic), dtype=np.int32)
    if prog.size < 4:
        raise RuntimeError('Invalid binary logic data')
    n_in, n_out = int(prog[2]), int(prog[3])
    x = np.ascontiguousarray(np.ravel(data), dtype=np.float64)
    assert n_in > 0 and x.size % n_in == 0, f'Input size {x.size} is not divisible by {n_in}'
    n_samples = x.size // n_in
    out = np.zeros((n_samples, max(n_out, 0)), dtype=np.float64)
    L = lib()
    L.da_dais_run_on.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int]
    L.da_dais_last_error.restype = C.c_char_p
    if executor not in _DAIS_EXECUTORS:
        raise ValueError(f'unknown DAIS executor {executor!r} (one of {sorted(_DAIS_EXECUTORS)})')
    if L.da_dais_run_on(prog.ctypes.data, prog.size, x.ctypes.data, n_samples, out.ctypes.data, int(n_threads), _DAIS_EXECUTORS[executor]) != 0:
        raise RuntimeError(L.da_dais_last_error().decode())
    return out


# the reference exposes the raw extension module as `da4ml._binary.cmvm_bin` (imported by trace/fixed_variable.py:15)
cmvm_bin = SimpleNamespace(
    solve=solve, kernel_decompose=kernel_decompose, csd_decompose=csd_decompose, int_arr_to_csd=int_arr_to_csd,
    get_lsb_loc=get_lsb_loc, iceil_log2=iceil_log2, cost_add=cost_add,
)  # fmt: skip

__all__ = ['dais_interp_run', 'int_arr_to_csd', 'csd_decompose', 'get_lsb_loc', 'kernel_decompose', 'solve', 'iceil_log2', 'solve_many', 'cost_add']
