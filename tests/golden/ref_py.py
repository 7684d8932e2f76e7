"""Import the reference's *Python* package in the build container (never on the GPU box, never from the tests).

The reference's ``da4ml._binary`` is a nanobind module that cannot be built here (oracle/README.md); its functions are
supplied by ``oracle/_ref/libref.so`` -- the reference's own C++ sources behind a C shim -- so that the reference's
pure-Python layers (``da4ml.types``, ``da4ml.trace.*``) run unmodified from ``/root/reference/src``.  Used only by the
``make_*_golden.py`` scripts.
"""

import sys
import types
from pathlib import Path

import numpy as np

REF_SRC = Path('/root/reference/src/da4ml')
ROOT = Path(__file__).resolve().parent.parent.parent


def load():
    """Returns the imported reference package modules (types, trace.pipeline, trace.tracer, trace.fixed_variable)."""
    if 'da4ml' in sys.modules and getattr(sys.modules['da4ml'], '_da4ml_amd_stub', False):
        pkg = sys.modules['da4ml']
        return pkg._mods
    if not REF_SRC.exists():
        raise FileNotFoundError(f'{REF_SRC} is only present in the build container')
    sys.path.insert(0, str(ROOT))
    from oracle.oracle import Oracle

    R = Oracle('ref')
    pkg = types.ModuleType('da4ml')
    pkg.__path__ = [str(REF_SRC)]  # the package's own __init__ (codegen, converters ...) is not run
    pkg._da4ml_amd_stub = True
    sys.modules['da4ml'] = pkg

    def _no(*a, **k):
        raise RuntimeError('not available through the libref stub')

    def cost_add(q0, q1, shift, sub, adder_size, carry_size):
        return R.cost_add(tuple(q0), tuple(q1), int(shift), bool(sub), int(adder_size), int(carry_size))

    binary = types.ModuleType('da4ml._binary')
    binary.__path__ = []
    cmvm_bin = types.ModuleType('da4ml._binary.cmvm_bin')
    cmvm_bin.cost_add = cost_add
    cmvm_bin.get_lsb_loc = lambda x: R.get_lsb_loc(float(x))
    cmvm_bin.iceil_log2 = lambda x: R.iceil_log2(float(x))
    for name in ('csd_decompose', 'int_arr_to_csd', 'kernel_decompose', 'solve'):
        setattr(cmvm_bin, name, _no)
    for name in ('csd_decompose', 'get_lsb_loc', 'iceil_log2', 'int_arr_to_csd', 'kernel_decompose', 'solve'):
        setattr(binary, name, getattr(cmvm_bin, name))
    binary.dais_interp_run = _no
    binary.cmvm_bin = cmvm_bin
    sys.modules['da4ml._binary'] = binary
    sys.modules['da4ml._binary.cmvm_bin'] = cmvm_bin

    # third-party import of the tracer's quantisation ops (HGQ's `quantizers`): absent here and not on this path
    for name in ('quantizers', 'quantizers.fixed_point', 'quantizers.fixed_point.fixed_point_ops_np'):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            mod.__path__ = []
            sys.modules[name] = mod
    sys.modules['quantizers.fixed_point.fixed_point_ops_np'].get_fixed_quantizer_np = _no

    import da4ml.types as rtypes  # noqa: E402
    import da4ml.trace.fixed_variable as rfv  # noqa: E402
    import da4ml.trace.pipeline as rpipe  # noqa: E402
    import da4ml.trace.tracer as rtracer  # noqa: E402

    pkg._mods = dict(types=rtypes, pipeline=rpipe, tracer=rtracer, fixed_variable=rfv, oracle=R)
    return pkg._mods


def to_ref_comb(rtypes, comb):
    """da4ml_amd CombLogic -> reference CombLogic (same fields)."""
    ops = [rtypes.Op(o.id0, o.id1, o.opcode, o.data, rtypes.QInterval(*o.qint), o.latency, o.cost) for o in comb.ops]
    return rtypes.CombLogic(tuple(comb.shape), list(comb.inp_shifts), list(comb.out_idxs), list(comb.out_shifts), list(comb.out_negs), ops,
                            comb.carry_size, comb.adder_size, None)  # fmt: skip


def to_ref_pipeline(rtypes, pipe):
    return rtypes.Pipeline(tuple(to_ref_comb(rtypes, s) for s in pipe.solutions))


def plain(obj):
    """Reference or local CombLogic / Pipeline -> nested plain lists (JSON-able), floats kept exactly."""
    if isinstance(obj, (np.integer,)):
        return int(obj)
    if isinstance(obj, (np.floating,)):
        return float(obj)
    if isinstance(obj, (np.bool_,)):
        return bool(obj)
    if isinstance(obj, (list, tuple)):
        return [plain(v) for v in obj]
    return obj
