"""Small host utilities with the reference's names and semantics
(``opty/utils.py``): only what the constraint/Jacobian path needs."""

import numpy as np

__all__ = ['parse_free', 'sort_sympy', 'coo_to_dense', 'ufuncify_matrix']


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


class _MatrixFunction(object):
    """``f(result, *num_args) -> result.reshape(n, rows, cols)``: the callable
    :func:`ufuncify_matrix` returns (``opty/utils.py:610-617``)."""

    def __init__(self, dag, outputs, num_vec, const_positions, num_args,
                 shape, tmp_dir=None, show_compile_output=False, device=0,
                 emit_options=None):
        from .codegen.program import matrix_program
        from .codegen.emit_hip import emit_matrix_module
        from . import hip_backend as hb
        self.shape = shape
        self.num_args = num_args
        self._on_torch_stream = False
        self.const_positions = tuple(const_positions)
        self.vec_positions = tuple(k for k in range(num_args)
                                   if k not in self.const_positions)
        assert len(self.vec_positions) == num_vec
        import os
        if os.environ.get('OPTY_COLLECT', '1') != '0':
            from .codegen.simplify import collect_coefficients
            outputs = collect_coefficients(dag, list(outputs))
        prog = matrix_program(dag, outputs, num_vec,
                              len(self.const_positions), shape)
        self.source, self.meta = emit_matrix_module(prog, emit_options)
        hsaco = hb.compile_module(self.source, tmp_dir, show_compile_output)
        k = self.meta['kernels']['jac']
        self._hip = hb.HipMatrix(dict(
            num_vec=num_vec, num_const=len(self.const_positions),
            rows=shape[0], cols=shape[1], wgs_per_block=k['wgs_per_block'],
            waves_per_wg=k['waves_per_wg'],
            num_uniform=self.meta['num_uniform'], device=int(device)), hsaco)

    @property
    def hip(self):
        """The :class:`opty_amd.hip_backend.HipMatrix` handle."""
        return self._hip

    def __call__(self, result, *num_args):
        from . import hip_backend as hb
        rows, cols = self.shape
        if len(num_args) != self.num_args:
            raise TypeError('expected %d arguments after result, got %d'
                            % (self.num_args, len(num_args)))
        on_device = hasattr(result, 'data_ptr')
        if on_device:
            ok = (result.dim() == 2 and result.is_contiguous() and
                  str(result.dtype) == 'torch.float64' and result.is_cuda)
        else:
            ok = (isinstance(result, np.ndarray) and result.ndim == 2 and
                  result.dtype == np.float64 and result.flags['C_CONTIGUOUS'])
        if not ok or result.shape[1] != rows*cols:
            raise ValueError('result must be a C-contiguous float64 array of '
                             'shape (n, %d)' % (rows*cols))
        n = result.shape[0]
        vec = []
        for k in self.vec_positions:
            v = num_args[k]
            if on_device:
                good = (hasattr(v, 'data_ptr') and v.dim() == 1 and
                        v.is_contiguous() and v.is_cuda and
                        str(v.dtype) == 'torch.float64')
            else:
                # what a Cython ``double[::1]`` argument accepts
                # (opty/utils.py:778-793)
                good = (isinstance(v, np.ndarray) and v.ndim == 1 and
                        v.dtype == np.float64 and v.strides[0] == 8)
            if not good or v.shape[0] != n:
                raise ValueError('argument %d must be a contiguous float64 '
                                 'array of shape (%d,)' % (k, n))
            vec.append(v)
        cst = [float(num_args[k]) for k in self.const_positions]
        if n > 0:
            if on_device:
                # Stream contract of the device path: the launch is enqueued
                # on torch's CURRENT stream of the result's device -- ordered
                # after whatever produced the argument tensors there and
                # before whatever reads `result` there -- and is asynchronous
                # like any torch operation.  (The handle's own stream is not
                # ordered with torch's: a caller would read stale inputs or an
                # unfinished result.)
                import torch
                self._hip.use_torch_stream(
                    torch.cuda.current_stream(result.device))
            elif self._on_torch_stream:
                self._hip.set_stream(None)      # host path: own stream, sync
            self._on_torch_stream = bool(on_device)
            self._hip.evaluate(result, vec, cst, n,
                               hb.DEVICE if on_device else hb.HOST)
        return result.reshape(n, rows, cols)


def ufuncify_matrix(args, expr, const=None, tmp_dir=None, parallel=False,
                    show_compile_output=False, device=0, emit_options=None):
    """Returns ``f(result, *num_args) -> result.reshape(n, rows, cols)`` that
    evaluates a matrix of expressions for ``n`` argument rows on the GPU: the
    reference's plugin entry point with the reference's call contract
    (``opty/utils.py:639-640``; contract ``:610-617``, ``:778-807``).

    ``args``: the symbols of ``expr`` in call order; ``expr``: a SymPy Matrix
    or the 2-tuple ``cse()`` returns for one (``(replacements,
    [reduced_matrix])``, ``:677-682``); ``const``: those of ``args`` that are
    passed as floats (one value per call) instead of ``(n,)`` arrays.
    ``result`` is the caller's C-contiguous float64 ``(n, rows*cols)`` array
    (NumPy: evaluated through PCIe, synchronous; a CUDA ``torch`` tensor with
    CUDA tensor arguments: evaluated in place, enqueued on torch's current
    stream of that device -- ordered with the torch operations around it and
    asynchronous like them).

    ``tmp_dir`` is the code-object cache directory, ``parallel`` is accepted
    and ignored (a launch is always parallel); ``device`` (HIP ordinal) and
    ``emit_options`` (printer knobs) are extras.  A build failure raises
    ``ImportError`` with the compiler's stderr (``:912-916``).  Symbol
    *names* never reach the generated code (expressions are lowered to an
    operation DAG), so names that break the reference's C --
    ``d_{badsym}``, ``if``, ``I`` (``opty/tests/test_utils.py:244-336``) --
    are all fine here.
    """
    import sympy as sm
    from .codegen import ir
    from .codegen.lower import Lowerer
    args = list(args)
    const = tuple(const or ())
    if len(set(args)) != len(args):
        raise ValueError('args must be unique')
    missing = [c for c in const if c not in args]
    if missing:
        raise ValueError('const symbols {} are not in args'.format(missing))
    if hasattr(expr, 'shape'):
        replacements, matrix = (), sm.ImmutableDenseMatrix(expr)
    else:                                       # output of cse()
        replacements, matrix = expr[0], sm.ImmutableDenseMatrix(expr[1][0])
    dag = ir.DAG()
    table, const_positions, nvec = {}, [], 0
    for k, a in enumerate(args):
        if a in const:
            table[a] = dag.input('par', len(const_positions))
            const_positions.append(k)
        else:
            table[a] = dag.input('cur', nvec)
            nvec += 1
    low = Lowerer(dag, table)
    # cse() also pulls out Boolean sub-expressions that several Piecewise
    # conditions share (``x7 = x1 < 1/2``): those are not values of the DAG,
    # they are put back where they are used
    from sympy.logic.boolalg import Boolean
    bools = {}
    for sym, sub in replacements:
        if bools:
            sub = sub.xreplace(bools)
        if isinstance(sub, Boolean):
            bools[sym] = sub
        else:
            low.sym[sym] = low.lower(sub)
    outputs = [low.lower(e.xreplace(bools) if bools else e)
               for e in matrix]                 # row-major
    return _MatrixFunction(dag, outputs, nvec, const_positions, len(args),
                           matrix.shape, tmp_dir, show_compile_output,
                           device, emit_options)
