"""The two graph passes that consume solver output on the way to RTL:
``to_pipeline`` / ``retime_pipeline`` (reference ``src/da4ml/trace/pipeline.py``) and ``dead_statement_elimination``
(reference ``src/da4ml/trace/tracer.py:178-211``).  The symbolic tracer itself (``FixedVariable`` & co.) is out of scope;
retiming is implemented for what the CMVM solver emits (input copies, add, subtract)."""

from .pipeline import retime_pipeline, to_pipeline
from .tracer import dead_statement_elimination

__all__ = ['to_pipeline', 'retime_pipeline', 'dead_statement_elimination']
