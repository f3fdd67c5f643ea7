"""Test-only NumPy interpreter for the codegen DAG.

Lets the CPU test-suite check the lowering / differentiation
(``opty_amd.codegen``) against the golden vectors without a GPU.  It is NOT a
backend: nothing in ``opty_amd`` imports it.
"""
import numpy as np

from opty_amd.codegen import ir

_UN = {'sqrt': np.sqrt, 'sin': np.sin, 'cos': np.cos, 'tan': np.tan,
       'exp': np.exp, 'log': np.log, 'abs': np.abs, 'sign': np.sign,
       'asin': np.arcsin, 'acos': np.arccos, 'atan': np.arctan,
       'sinh': np.sinh, 'cosh': np.cosh, 'tanh': np.tanh,
       'step': lambda x: (x > 0).astype(float),
       'floor': np.floor, 'ceil': np.ceil, 'asinh': np.arcsinh,
       'acosh': np.arccosh, 'atanh': np.arctanh, 'log1p': np.log1p,
       'expm1': np.expm1, 'log2': np.log2, 'log10': np.log10,
       'exp2': np.exp2, 'cbrt': np.cbrt}
try:
    from scipy.special import erf as _erf, erfc as _erfc, gamma as _gamma, \
        gammaln as _gammaln
    _UN.update(erf=_erf, erfc=_erfc, tgamma=_gamma, lgamma=_gammaln)
except ImportError:                     # pragma: no cover
    import math
    _UN.update(erf=np.vectorize(math.erf), erfc=np.vectorize(math.erfc),
               tgamma=np.vectorize(math.gamma),
               lgamma=np.vectorize(math.lgamma))
_REL = {'lt': np.less, 'le': np.less_equal, 'eq': np.equal,
        'ne': np.not_equal}


def evaluate(dag, roots, inputs):
    """``inputs(kind, index)`` -> scalar or (nodes,) array."""
    val = {}
    for i in dag.reachable(roots):
        op, a = dag.op[i], dag.args[i]
        if op == ir.CONST:
            v = a[0]
        elif op == ir.INPUT:
            v = inputs(*a)
        elif op == ir.ADD:
            v = val[a[0]] + val[a[1]]
        elif op == ir.SUB:
            v = val[a[0]] - val[a[1]]
        elif op == ir.MUL:
            v = val[a[0]]*val[a[1]]
        elif op == ir.DIV:
            v = val[a[0]]/val[a[1]]
        elif op == ir.NEG:
            v = -val[a[0]]
        elif op == ir.POWI:
            v = val[a[0]]**a[1]
        elif op == ir.POW:
            v = np.power(val[a[0]], val[a[1]])
        elif op == ir.MAX:
            v = np.maximum(val[a[0]], val[a[1]])
        elif op == ir.MIN:
            v = np.minimum(val[a[0]], val[a[1]])
        elif op == ir.ATAN2:
            v = np.arctan2(val[a[0]], val[a[1]])
        elif op == ir.SELECT:
            v = np.where(_REL[a[0]](val[a[1]], val[a[2]]), val[a[3]],
                         val[a[4]])
        else:
            v = _UN[op](np.asarray(val[a[0]], dtype=float))
        val[i] = v
    return [val[r] for r in roots]


# d f / d x of the unary functions, for the error propagation below
_DUN = {'sqrt': lambda x, f: 0.5/f, 'sin': lambda x, f: np.cos(x),
        'cos': lambda x, f: np.sin(x), 'tan': lambda x, f: 1.0 + f*f,
        'exp': lambda x, f: f, 'log': lambda x, f: 1.0/x,
        'abs': lambda x, f: 1.0, 'sign': lambda x, f: 0.0,
        'asin': lambda x, f: 1.0/np.sqrt(1.0 - x*x),
        'acos': lambda x, f: 1.0/np.sqrt(1.0 - x*x),
        'atan': lambda x, f: 1.0/(1.0 + x*x), 'sinh': lambda x, f: np.cosh(x),
        'cosh': lambda x, f: np.sinh(x), 'tanh': lambda x, f: 1.0 - f*f,
        'step': lambda x, f: 0.0, 'floor': lambda x, f: 0.0,
        'ceil': lambda x, f: 0.0,
        'erf': lambda x, f: 1.1283791670955126*np.exp(-x*x),
        'erfc': lambda x, f: 1.1283791670955126*np.exp(-x*x),
        'asinh': lambda x, f: 1.0/np.sqrt(x*x + 1.0),
        'acosh': lambda x, f: 1.0/np.sqrt(x*x - 1.0),
        'atanh': lambda x, f: 1.0/(1.0 - x*x),
        'log1p': lambda x, f: 1.0/(1.0 + x), 'expm1': lambda x, f: f + 1.0,
        'log2': lambda x, f: 1.4426950408889634/x,
        'log10': lambda x, f: 0.4342944819032518/x,
        'exp2': lambda x, f: 0.6931471805599453*f,
        'cbrt': lambda x, f: 1.0/(3.0*f*f),
        'tgamma': lambda x, f: f*_digamma(x),
        'lgamma': lambda x, f: _digamma(x)}


def _digamma(x):
    from scipy.special import digamma
    return digamma(x)


def evaluate_with_error_bound(dag, roots, inputs):
    """Values and first-order rounding-error bounds, in units of the float64
    unit round-off: ``|computed - exact| <~ u * bound`` for ANY evaluation
    order of the same sums and products (running error analysis: every
    operation contributes one rounding of its own result plus its operands'
    errors scaled by the partial derivatives).  An entry that is a sum of
    large cancelling terms gets a bound at the size of the terms, an entry
    that is a constant or a single product a bound at its own size -- the
    per-entry floor of the parity tolerance (``golden_util.assert_close``).
    """
    val, err = {}, {}
    with np.errstate(all='ignore'):
        for i in dag.reachable(roots):
            op, a = dag.op[i], dag.args[i]
            if op == ir.CONST:
                v, e = a[0], 0.0
            elif op == ir.INPUT:
                v, e = inputs(*a), 0.0
            elif op in (ir.ADD, ir.SUB):
                v = val[a[0]] + val[a[1]] if op == ir.ADD \
                    else val[a[0]] - val[a[1]]
                e = err[a[0]] + err[a[1]]
            elif op == ir.MUL:
                x, y = val[a[0]], val[a[1]]
                v = x*y
                e = np.abs(y)*err[a[0]] + np.abs(x)*err[a[1]]
            elif op == ir.DIV:
                x, y = val[a[0]], val[a[1]]
                v = x/y
                e = err[a[0]]/np.abs(y) + np.abs(v/y)*err[a[1]]
            elif op == ir.NEG:
                v, e = -val[a[0]], err[a[0]]
            elif op == ir.POWI:
                x = val[a[0]]
                v = x**a[1]
                e = a[1]*np.abs(x**(a[1] - 1))*err[a[0]]
            elif op == ir.POW:
                x, y = val[a[0]], val[a[1]]
                v = np.power(x, y)
                e = np.abs(v*y/x)*err[a[0]] + np.abs(v*np.log(np.abs(x)))*err[a[1]]
            elif op in (ir.MAX, ir.MIN):
                pick = np.maximum if op == ir.MAX else np.minimum
                v = pick(val[a[0]], val[a[1]])
                e = np.maximum(err[a[0]], err[a[1]])
            elif op == ir.ATAN2:
                y, x = val[a[0]], val[a[1]]
                v = np.arctan2(y, x)
                r2 = x*x + y*y
                e = (np.abs(x)*err[a[0]] + np.abs(y)*err[a[1]])/r2
            elif op == ir.SELECT:
                c = _REL[a[0]](val[a[1]], val[a[2]])
                v = np.where(c, val[a[3]], val[a[4]])
                e = np.where(c, err[a[3]], err[a[4]])
            else:
                x = np.asarray(val[a[0]], dtype=float)
                v = _UN[op](x)
                e = np.abs(_DUN[op](x, v))*err[a[0]]
            val[i] = v
            # one rounding of the operation's own result (2 for libm calls)
            err[i] = e + np.abs(v)*(2.0 if op in _UN or op in (
                ir.POW, ir.ATAN2) else (0.0 if op in (
                    ir.CONST, ir.INPUT, ir.NEG, ir.SELECT) else 1.0))
    return [val[r] for r in roots], [err[r] for r in roots]


def error_bounds(col, free, nodes=None):
    """Per-entry rounding-error bounds (units of round-off) of
    ``constraints(free)`` and ``jacobian(free)`` in the reference's layouts;
    ``nodes``: only these constraint nodes -> ``con (M, len(nodes))``,
    ``jac (len(nodes), P)`` plus the instance tails."""
    prog = col._build_program()
    inputs, count = _input_getter(col, free, nodes)
    ones = np.ones(count)
    _, ce = evaluate_with_error_bound(prog.dag, prog.con_out, inputs)
    _, je = evaluate_with_error_bound(prog.dag, prog.jac_out, inputs)
    _, ice = evaluate_with_error_bound(prog.dag, prog.inst_con_out, inputs)
    _, ije = evaluate_with_error_bound(prog.dag, prog.inst_jac_out, inputs)
    con = np.stack([np.atleast_1d(c)*ones for c in ce]) if ce \
        else np.zeros((0, count))
    jac = np.stack([np.atleast_1d(v)*ones for v in je], axis=1) if je \
        else np.zeros((count, 0))
    ic, ij = np.array(ice, dtype=float), np.array(ije, dtype=float)
    if nodes is None:
        return np.concatenate((con.ravel(), ic)), \
            np.concatenate((jac.ravel(), ij))
    return con, jac, ic, ij


def _input_getter(col, free, nodes=None):
    prog = col._build_program()
    N, n, q = col.num_collocation_nodes, prog.n, prog.q
    free = np.asarray(free, dtype=float)
    known = np.array([col.known_trajectory_map[f](free)
                      if callable(col.known_trajectory_map[f])
                      else col.known_trajectory_map[f]
                      for f in col.known_input_trajectories], dtype=float)
    tail = free[(n + q)*N:]
    kpar = [float(col.known_parameter_map[p]) for p in col.known_parameters]
    sel = slice(None) if nodes is None else np.asarray(nodes)

    def row(r):
        src, k = prog.rows[r]
        return free[k*N:(k + 1)*N] if src == 'free' else known[k]

    def inputs(kind, idx):
        if kind in ('cur', 'adj'):
            off = prog.cur_offset if kind == 'cur' else prog.adj_offset
            return row(idx)[off:off + N - 1][sel]
        if kind == 'par':
            src, k = prog.pars[idx]
            return kpar[k] if src == 'known' else tail[k]
        if kind == 'h':
            return col.node_time_interval if prog.h[0] == 'fixed' \
                else tail[prog.h[1]]
        if kind == 'free':
            f = col._inst_atoms[idx]
            return free[col.instance_constraints_free_index_map[f]]
        raise AssertionError(kind)

    return inputs, (N - 1 if nodes is None else len(sel))


def evaluate_collocator(col, free):
    """constraints(free), jacobian(free) of an ``opty_amd.ConstraintCollocator``
    through the interpreter (layouts as the reference's)."""
    prog = col._build_program()
    N, n, q = col.num_collocation_nodes, prog.n, prog.q
    free = np.asarray(free, dtype=float)
    known = np.array([col.known_trajectory_map[f](free)
                      if callable(col.known_trajectory_map[f])
                      else col.known_trajectory_map[f]
                      for f in col.known_input_trajectories], dtype=float)
    tail = free[(n + q)*N:]
    kpar = [float(col.known_parameter_map[p]) for p in col.known_parameters]

    def row(r):
        src, k = prog.rows[r]
        return free[k*N:(k + 1)*N] if src == 'free' else known[k]

    def inputs(kind, idx):
        if kind in ('cur', 'adj'):
            off = prog.cur_offset if kind == 'cur' else prog.adj_offset
            return row(idx)[off:off + N - 1]
        if kind == 'par':
            src, k = prog.pars[idx]
            return kpar[k] if src == 'known' else tail[k]
        if kind == 'h':
            return col.node_time_interval if prog.h[0] == 'fixed' \
                else tail[prog.h[1]]
        if kind == 'free':
            f = col._inst_atoms[idx]
            return free[col.instance_constraints_free_index_map[f]]
        raise AssertionError(kind)

    ones = np.ones(N - 1)
    con = evaluate(prog.dag, prog.con_out, inputs)
    con = np.concatenate([np.atleast_1d(c)*ones for c in con]) \
        if con else np.zeros(0)
    jac = evaluate(prog.dag, prog.jac_out, inputs)
    jac = np.stack([np.atleast_1d(v)*ones for v in jac], axis=1).ravel()
    ic = evaluate(prog.dag, prog.inst_con_out, inputs)
    ij = evaluate(prog.dag, prog.inst_jac_out, inputs)
    con = np.concatenate((con, np.array(ic, dtype=float)))
    jac = np.concatenate((jac, np.array(ij, dtype=float)))
    return con, jac
