"""Small host utilities with the reference's names and semantics
(``opty/utils.py``): only what the constraint/Jacobian path needs."""

import numpy as np

__all__ = ['parse_free', 'sort_sympy', 'coo_to_dense']


def sort_sympy(seq):
    """Symbols sorted by name, applied functions by their class name
    (``opty/utils.py:473-480``)."""
    seq = list(seq)
    try:
        return sorted(seq, key=lambda x: x.name)
    except AttributeError:
        return sorted(seq, key=lambda x: x.__class__.__name__)


def parse_free(free, n, q, N, variable_duration=False):
    """Views into the free vector (``opty/utils.py:277-326``).

    Returns ``states (n, N)``, ``specified`` (``None`` if q == 0, ``(N,)`` if
    q == 1, else ``(q, N)``), ``constants (r,)`` and, if
    ``variable_duration``, the node time interval ``free[-1]``.
    """
    states = free[:n*N].reshape((n, N))
    if q == 0:
        specified = None
    else:
        specified = free[n*N:(n + q)*N]
        if q > 1:
            specified = specified.reshape((q, N))
    if variable_duration:
        return states, specified, free[(n + q)*N:-1], free[-1]
    return states, specified, free[(n + q)*N:]


def coo_to_dense(values, rows, cols):
    """Dense matrix from COO triplets where the LAST duplicate wins, the
    semantics of the reference's ``_coo_matrix`` (``opty/utils.py:38-44``)
    that every Jacobian test of the reference relies on."""
    out = np.zeros((int(rows.max()) + 1, int(cols.max()) + 1),
                   dtype=values.dtype)
    out[rows, cols] = values
    return out
