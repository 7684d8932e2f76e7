"""Public CMVM surface -- mirrors the reference's ``da4ml.cmvm`` (src/da4ml/cmvm/__init__.py:7-29)."""

from collections.abc import Callable
from typing import TypedDict

import numpy as np

from .._binary import kernel_decompose, solve, solve_many
from ..types import CombLogic, Op, QInterval


class solver_options_t(TypedDict, total=False):
    method0: str
    method1: str
    hard_dc: int
    decompose_dc: int
    adder_size: int
    carry_size: int
    search_all_decompose_dc: bool
    offload_fn: None | Callable[[np.ndarray, object], np.ndarray]


__all__ = ['solve', 'solve_many', 'QInterval', 'Op', 'CombLogic', 'kernel_decompose', 'solver_options_t']
