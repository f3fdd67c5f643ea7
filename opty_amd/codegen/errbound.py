"""Running rounding-error analysis of the expression DAG (NumPy, host).

Used by the build referee (``ConstraintCollocator._verify_build``) to settle
a MARGINAL disagreement between a compiled kernel and the instruction tape:
both evaluate the same DAG, so whatever the compiler reorders or contracts
into FMAs, an entry may differ from the tape's by a small multiple of the
first-order rounding-error bound of ITS OWN operations -- and by no more.  An
ill-conditioned entry (a sum of large cancelling terms) gets a bound the size
of its terms, a constant or a single product a bound of its own size; a build
that is off by more than that is a compiler fault however small the number
looks next to the row's largest value.  The parity tests use the same
analysis for their per-entry tolerance floors (``tests/dag_interp.py``).

Nothing here supplies a value a caller sees: host libm, test/referee use only.
"""
import numpy as np

from . import ir

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
