"""opty_amd -- MI355X-native constraint and Jacobian evaluator for opty-style
direct collocation (drop-in for ``opty.ConstraintCollocator`` / ``Problem``
on the hot path)."""

from .direct_collocation import (ConstraintCollocator, Problem,
                                 ShardedProblem)
from .utils import parse_free, ufuncify_matrix
from .objective import create_objective_function

__all__ = ['ConstraintCollocator', 'Problem', 'ShardedProblem', 'parse_free',
           'ufuncify_matrix',
           'create_objective_function']
__version__ = '0.1.0'
