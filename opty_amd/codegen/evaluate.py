"""Host-side evaluation of node-invariant DAG nodes.

``ConstraintCollocator(specialize_parameters=True)`` prints the
node-invariant sub-expressions of a problem -- products of masses and
lengths, ``1/h`` ... -- as float64 LITERALS of the generated module instead of
reading them from the table ``opty_uni`` fills: their values are computed here,
once per build, from the known parameter values and the fixed node time
interval, one individually rounded operation per DAG node (the same operations
the device kernel performs; the library functions are the host's ``libm``
instead of the device's, i.e. equal to rounding).
"""

import math

from . import ir


def _sign(x):
    return float((x > 0.0) - (x < 0.0))


_UNARY = {
    'sqrt': math.sqrt, 'sin': math.sin, 'cos': math.cos, 'tan': math.tan,
    'exp': math.exp, 'log': math.log, 'abs': abs, 'sign': _sign,
    'asin': math.asin, 'acos': math.acos, 'atan': math.atan,
    'sinh': math.sinh, 'cosh': math.cosh, 'tanh': math.tanh,
    'step': lambda x: 1.0 if x > 0.0 else 0.0, 'erf': math.erf,
    'erfc': math.erfc, 'floor': lambda x: float(math.floor(x)),
    'ceil': lambda x: float(math.ceil(x)), 'asinh': math.asinh,
    'acosh': math.acosh, 'atanh': math.atanh, 'log1p': math.log1p,
    'expm1': math.expm1, 'log2': math.log2, 'log10': math.log10,
    'exp2': lambda x: 2.0**x,
    'cbrt': lambda x: math.copysign(abs(x)**(1.0/3.0), x),
    'tgamma': math.gamma, 'lgamma': math.lgamma}

_REL = {'lt': lambda a, b: a < b, 'le': lambda a, b: a <= b,
        'eq': lambda a, b: a == b, 'ne': lambda a, b: a != b}


def _powi(x, n):
    """``x**n`` with the multiplications the printer emits: ``x*x``,
    ``x*x*x``, then ``opty_powi<n>`` of opty_device.h (halving)."""
    if n == 2:
        return x*x
    if n == 3:
        return x*x*x
    if n == 1:
        return x
    if n % 2 == 0:
        y = _powi_dev(x, n//2)
        return y*y
    return x*_powi_dev(x, n - 1)


def _powi_dev(x, n):
    if n == 1:
        return x
    if n % 2 == 0:
        y = _powi_dev(x, n//2)
        return y*y
    return x*_powi_dev(x, n - 1)


def evaluate_uniform(dag, roots, scalar):
    """``{node: float}`` for ``roots`` and everything below them.

    ``scalar(kind, index)``: value of a node-invariant INPUT node (``'par'``,
    ``'h'``), or None when it is not known at build time (a value that lives
    in ``free``): nodes that depend on such an input are left out.  Domain
    errors give NaN / inf like the device's functions."""
    vals = {}
    unknown = set()
    for i in dag.reachable(roots):
        op, args = dag.op[i], dag.args[i]
        if op == ir.CONST:
            vals[i] = dag.value(i)
            continue
        if op == ir.INPUT:
            v = scalar(*args) if dag.uni[i] else None
            if v is None:
                unknown.add(i)
            else:
                vals[i] = float(v)
            continue
        ops = dag.operands(i)
        if any(j in unknown for j in ops):
            unknown.add(i)
            continue
        a = [vals[j] for j in ops]
        try:
            if op == ir.ADD:
                v = a[0] + a[1]
            elif op == ir.SUB:
                v = a[0] - a[1]
            elif op == ir.MUL:
                v = a[0]*a[1]
            elif op == ir.DIV:
                v = a[0]/a[1] if a[1] != 0.0 else (
                    float('nan') if a[0] == 0.0 or a[0] != a[0]
                    else math.copysign(float('inf'), a[0]) *
                    math.copysign(1.0, a[1]))
            elif op == ir.NEG:
                v = -a[0]
            elif op == ir.POWI:
                v = _powi(a[0], args[1])
            elif op == ir.POW:
                v = math.pow(a[0], a[1])
            elif op == ir.MAX:
                v = max(a[0], a[1])
            elif op == ir.MIN:
                v = min(a[0], a[1])
            elif op == ir.ATAN2:
                v = math.atan2(a[0], a[1])
            elif op == ir.SELECT:
                v = a[2] if _REL[args[0]](a[0], a[1]) else a[3]
            else:
                v = _UNARY[op](a[0])
        except (ValueError, ZeroDivisionError, OverflowError) as exc:
            v = _ieee_result(op, a, exc)
        vals[i] = float(v)
    return vals


def _ieee_result(op, a, exc):
    """What C's libm (and the device's) returns where Python raises: a pole
    gives a SIGNED infinity (``log(0) = -inf``, ``pow(0, -1) = +inf``,
    ``atanh(-1) = -inf``, ``lgamma`` / ``tgamma`` at their poles), an overflow
    the infinity of the result's sign (``sinh(-800) = -inf``, ``pow(-10, 301)
    = -inf``), a domain error NaN (every overflow used to map to
    +inf and every ValueError to NaN)."""
    inf, nan = float('inf'), float('nan')
    x = a[0]
    if any(v != v for v in a):
        return nan
    if op in ('log', 'log2', 'log10'):
        return -inf if x == 0.0 else nan
    if op == 'log1p':
        return -inf if x == -1.0 else nan
    if op == 'atanh':
        return math.copysign(inf, x) if abs(x) == 1.0 else nan
    if op in ('exp', 'exp2', 'expm1', 'cosh'):
        return inf if isinstance(exc, OverflowError) else nan
    if op == 'sinh':
        return math.copysign(inf, x) if isinstance(exc, OverflowError) \
            else nan
    if op == 'tgamma':
        if isinstance(exc, OverflowError):
            return inf
        if x == 0.0:
            return math.copysign(inf, x)
        return nan                      # negative integers: C gives NaN
    if op == 'lgamma':
        return inf                      # poles at 0, -1, -2 ...: +inf
    if op in (ir.POW, ir.POWI):
        b = a[1] if op == ir.POW else None
        if op == ir.POWI:
            # (Python floats do not raise on multiplication: unreachable,
            # kept for completeness)
            return inf
        if x == 0.0 and b < 0.0:
            # pow(+-0, negative): +-inf for odd integers, +inf otherwise
            odd = b == int(b) and int(b) % 2 != 0
            return math.copysign(inf, x) if odd else inf
        if isinstance(exc, OverflowError):
            odd = b == int(b) and int(b) % 2 != 0
            return -inf if (x < 0.0 and odd) else inf
        return nan                      # negative base, fractional exponent
    if isinstance(exc, OverflowError):
        return inf
    return nan

