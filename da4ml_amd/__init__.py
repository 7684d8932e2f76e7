"""da4ml_amd -- MI355X-native drop-in for the CMVM optimiser path of calad0i/da4ml.

Only the hot path is provided: ``da4ml_amd.cmvm.solve`` / ``kernel_decompose`` and the
helpers of ``da4ml_amd._binary``, with the result types of ``da4ml_amd.types``.  All compute goes through
``libda4ml_hip.so`` (hand-written HIP kernels for gfx950); there is no CPU fallback.
"""

from . import types  # noqa: F401
from .types import CombLogic, Op, Pipeline, QInterval  # noqa: F401

__version__ = '0.1.0'


def install_as_da4ml():
    """Alias this package as ``da4ml`` in ``sys.modules`` so that ``from da4ml.cmvm import solve`` resolves here.

    Only the CMVM path exists (plus the graph passes ``da4ml.trace.to_pipeline`` / ``dead_statement_elimination`` that
    consume its output); the symbolic tracer and the codegen parts of da4ml are not provided.
    """
    import sys

    from . import _binary, cmvm, trace, typing
    from .trace import pipeline as trace_pipeline
    from .trace import tracer as trace_tracer

    me = sys.modules[__name__]
    for name, mod in (('da4ml', me), ('da4ml.types', types), ('da4ml._binary', _binary), ('da4ml.cmvm', cmvm), ('da4ml.typing', typing),
                      ('da4ml.trace', trace), ('da4ml.trace.pipeline', trace_pipeline), ('da4ml.trace.tracer', trace_tracer)):
        sys.modules.setdefault(name, mod)
