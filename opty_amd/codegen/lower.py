"""SymPy -> DAG lowering and forward-mode differentiation on the DAG.

Replaces, for the HIP backend, the reference's ``sm.cse`` + C printing
(``opty/utils.py:745-757``) and its symbolic ``_forward_jacobian``
(``opty/utils.py:82-228``): the structural sharing comes from hash-consing in
:class:`opty_amd.codegen.ir.DAG`, the Jacobian from a sparse forward sweep that
carries, for every DAG node, a ``{wrt-column: derivative node}`` dictionary.
"""

import sympy as sm

from . import ir


class LoweringError(NotImplementedError):
    pass


_UNARY_FUNCS = {
    sm.sin: 'sin', sm.cos: 'cos', sm.tan: 'tan', sm.exp: 'exp',
    sm.log: 'log', sm.Abs: 'abs', sm.sign: 'sign', sm.asin: 'asin',
    sm.acos: 'acos', sm.atan: 'atan', sm.sinh: 'sinh', sm.cosh: 'cosh',
    sm.tanh: 'tanh', sm.erf: 'erf', sm.erfc: 'erfc', sm.floor: 'floor',
    sm.ceiling: 'ceil', sm.asinh: 'asinh', sm.acosh: 'acosh',
    sm.atanh: 'atanh',
}
try:
    from sympy.codegen import cfunctions as _cf
    _UNARY_FUNCS.update({_cf.log1p: 'log1p', _cf.expm1: 'expm1',
                         _cf.log2: 'log2', _cf.log10: 'log10',
                         _cf.exp2: 'exp2', _cf.Cbrt: 'cbrt',
                         _cf.Sqrt: 'sqrt'})
except ImportError:                     # pragma: no cover
    _cf = None
_UNARY_FUNCS.update({sm.gamma: 'tgamma', sm.loggamma: 'lgamma'})
#: printed by the reference's C99 printer as 1/cos, 1/sin, cos/sin ...
_RECIPROCAL_FUNCS = {sm.sec: 'cos', sm.csc: 'sin', sm.sech: 'cosh',
                     sm.csch: 'sinh'}


def _c99_rewrites():
    """``{function name: target function}`` of the C99 printer's "simple
    rewrite to a supported function" table (``CodePrinter.
    _rewriteable_functions``; the reference's printer inherits it,
    ``opty/utils.py:61``)."""
    try:
        from sympy.printing.c import C99CodePrinter
        table = C99CodePrinter._rewriteable_functions
    except (ImportError, AttributeError):       # pragma: no cover
        return {}
    out = {}
    for name, (target, _) in table.items():
        fn = getattr(sm, target, None)
        if fn is not None:
            out[name] = fn
    return out


_C99_REWRITES = _c99_rewrites()


def _printer_expansion(e):
    """The expression the reference's code printer prints *in place of* an
    applied function that is not in its function table, or None.

    The reference prints every sub-expression with a ``C99CodePrinter``
    (``opty/utils.py:61-79``, ``:751-757``), whose dispatch accepts far more
    than the table of C math functions:

    1. an object's own printer hook (``_ccode``) wins -- the musculotendon
       curves of ``sympy.physics.biomechanics`` print their defining
       expression, ``self.doit(deep=False, evaluate=False)``;
    2. a function with a numeric implementation attached as a ``Lambda``
       (``implemented_function``) is inlined;
    3. functions in the printer's rewrite table (``acot``, ``asec``,
       ``binomial``, ``frac``, ``SingularityFunction`` ...) are rewritten in
       terms of a printable one.

    The HIP backend lowers that same expansion into the DAG; derivatives then
    come from forward mode over the expansion, where the reference
    differentiates symbolically through ``fdiff`` -- equal functions.  An
    undefined function (``r(x)`` that is not a known trajectory) has no
    expansion, here as there.
    """
    if not isinstance(e, sm.Function) or \
            isinstance(e, sm.core.function.AppliedUndef) and \
            not hasattr(e, '_imp_'):
        return None
    cands = []
    if hasattr(e, '_ccode') or hasattr(e, '_print_code'):
        for kw in (dict(deep=False, evaluate=False), dict(deep=False)):
            try:
                cands.append(e.doit(**kw))
                break
            except TypeError:
                continue
    imp = getattr(e, '_imp_', None)
    if isinstance(imp, sm.Lambda):
        cands.append(imp(*e.args))
    target = _C99_REWRITES.get(e.func.__name__)
    if target is not None:
        cands.append(e.rewrite(target))
    if not cands:
        # last resort, also what ``_print_Function`` falls back on for
        # subclasses that evaluate lazily
        try:
            cands.append(e.doit(deep=False))
        except Exception:                       # noqa: BLE001
            pass
    for x in cands:
        if isinstance(x, sm.Basic) and x != e and not x.has(e.func):
            return x
    return None


class Lowerer(object):
    """Lowers SymPy expressions over a fixed symbol table into one DAG."""

    def __init__(self, dag, symbol_nodes):
        self.dag = dag
        self.sym = dict(symbol_nodes)     # sympy Symbol -> node id
        self._memo = {}

    def lower(self, expr):
        expr = sm.sympify(expr)
        # explicit stack: (expr, visited) post-order, memoised on the SymPy
        # expression so shared sub-trees are lowered once.
        memo = self._memo
        stack = [(expr, False)]
        while stack:
            e, ready = stack.pop()
            if e in memo:
                continue
            if e in self.sym:
                # a whole expression bound to an input (e.g. the applied
                # function r_i(x_i) of an implicit known trajectory)
                memo[e] = self.sym[e]
                continue
            if not ready and e.args and not e.is_Number and \
                    not isinstance(e, sm.Piecewise):
                stack.append((e, True))
                for a in e.args:
                    if a not in memo:
                        stack.append((a, False))
                continue
            memo[e] = self._convert(e)
        return memo[expr]

    def _convert(self, e):
        d = self.dag
        m = self._memo
        if e.is_Symbol:
            try:
                return self.sym[e]
            except KeyError:
                raise LoweringError('symbol %s is not an argument of the '
                                    'discretised equations' % e)
        if e.is_Number or isinstance(e, sm.NumberSymbol):
            return d.const(float(e))
        if e.is_Add:
            terms = [m[a] for a in e.args]
            # node-invariant terms first so that they pair up with each other
            terms.sort(key=lambda i: (not d.uni[i],))
            return d.sum(terms)
        if e.is_Mul:
            num_u, num_v, den = [], [], []
            for a in e.args:
                if a.is_Pow and a.exp.is_Number and a.exp.is_negative:
                    den.append(m[a.base] if a.exp == -1 else
                               self._pow(m[a.base], -a.exp))
                else:
                    (num_u if d.uni[m[a]] else num_v).append(m[a])
            num = d.mul(d.prod(num_u), d.prod(num_v)) if num_u else \
                d.prod(num_v)
            if den:
                # a node-invariant denominator (typically h) becomes one
                # shared reciprocal and a multiplication: x/h -> x*(1/h)
                den_u = [i for i in den if d.uni[i]]
                den_v = [i for i in den if not d.uni[i]]
                if den_u:
                    num = d.mul(d.div(d.one, d.prod(den_u)), num)
                if den_v:
                    num = d.div(num, d.prod(den_v))
            return num
        if e.is_Pow:
            return self._pow(m[e.base], e.exp, m.get(e.exp))
        f = e.func
        if f in _UNARY_FUNCS:
            return d.unary(_UNARY_FUNCS[f], m[e.args[0]])
        if f is sm.atan2:
            return d.binary(ir.ATAN2, m[e.args[0]], m[e.args[1]])
        if f in (sm.Max, sm.Min):
            name = ir.MAX if f is sm.Max else ir.MIN
            acc = m[e.args[0]]
            for a in e.args[1:]:
                acc = d.binary(name, acc, m[a])
            return acc
        if _cf is not None and f is _cf.hypot:
            a, b = m[e.args[0]], m[e.args[1]]
            return d.unary('sqrt', d.add(d.mul(a, a), d.mul(b, b)))
        if _cf is not None and f is _cf.fma:
            return d.add(d.mul(m[e.args[0]], m[e.args[1]]), m[e.args[2]])
        if f is sm.sinc:
            # the C printer's ((x != 0) ? sin(x)/x : 1)
            x = m[e.args[0]]
            return d.select('ne', x, d.zero, d.div(d.unary('sin', x), x),
                            d.one)
        if f is sm.Heaviside:
            return d.unary('step', m[e.args[0]])
        if f in _RECIPROCAL_FUNCS:
            return d.div(d.one, d.unary(_RECIPROCAL_FUNCS[f], m[e.args[0]]))
        if f is sm.cot:
            return d.div(d.unary('cos', m[e.args[0]]),
                         d.unary('sin', m[e.args[0]]))
        if f is sm.coth:
            return d.div(d.unary('cosh', m[e.args[0]]),
                         d.unary('sinh', m[e.args[0]]))
        if f is sm.Mod:
            # SymPy's Mod is the floored modulo: a - b*floor(a/b)
            a, b = m[e.args[0]], m[e.args[1]]
            return d.sub(a, d.mul(b, d.unary('floor', d.div(a, b))))
        if isinstance(e, sm.Piecewise):
            # ((e1, c1), (e2, c2), ...): the first true condition wins; no
            # true condition is undefined (NaN, as the C printer's code)
            acc = d.const(float('nan'))
            for expr, cond in reversed(e.args):
                acc = self._select(cond, self.lower(expr), acc)
            return acc
        if isinstance(e, sm.UnevaluatedExpr):
            return m[e.args[0]]
        expansion = _printer_expansion(e)
        if expansion is not None:
            return self.lower(expansion)
        raise LoweringError('cannot lower %s (%s) to the HIP backend'
                            % (f, e))

    def _select(self, cond, x, y):
        """``cond ? x : y`` for a SymPy Boolean over relationals."""
        d = self.dag
        if cond is sm.true or cond is True:
            return x
        if cond is sm.false or cond is False:
            return y
        if isinstance(cond, sm.And):
            acc = x
            for c in reversed(cond.args):
                acc = self._select(c, acc, y)
            return acc
        if isinstance(cond, sm.Or):
            acc = y
            for c in reversed(cond.args):
                acc = self._select(c, x, acc)
            return acc
        if isinstance(cond, sm.Not):
            return self._select(cond.args[0], y, x)
        if isinstance(cond, sm.ITE):
            # what SymPy folds a Piecewise nested in a Piecewise into:
            # ITE(a, b, c) ? x : y  ==  a ? (b ? x : y) : (c ? x : y)
            a, b, c = cond.args
            return self._select(a, self._select(b, x, y),
                                self._select(c, x, y))
        rels = {sm.Lt: ('lt', False), sm.Le: ('le', False),
                sm.Gt: ('lt', True), sm.Ge: ('le', True),
                sm.Eq: ('eq', False), sm.Ne: ('ne', False)}
        for cls, (rel, swap) in rels.items():
            if isinstance(cond, cls):
                a, b = self.lower(cond.lhs), self.lower(cond.rhs)
                if swap:
                    a, b = b, a
                return d.select(rel, a, b, x, y)
        raise LoweringError('cannot lower the condition %s' % (cond,))

    def _pow(self, base, exp, exp_node=None):
        d = self.dag
        exp = sm.sympify(exp)
        if exp.is_Integer:
            return d.powi(base, int(exp))
        if exp.is_Number:
            return d.pow(base, d.const(float(exp)))
        if exp_node is None:
            exp_node = self.lower(exp)
        return d.pow(base, exp_node)


def forward_jacobian(dag, outputs, wrt_inputs, chain=None):
    """Sparse forward-mode Jacobian on the DAG.

    ``outputs``: node ids of the M expressions; ``wrt_inputs``: node ids (INPUT
    nodes) of the C differentiation variables in column order.  ``chain``:
    ``{input node: [(wrt input node, derivative node)]}`` for inputs that are
    themselves known functions of a differentiation variable.  Returns an
    ``M x C`` list of lists of node ids (``dag.zero`` for structural zeros).
    """
    chain = chain or {}
    d = dag
    col_of = {node: k for k, node in enumerate(wrt_inputs)}
    order = d.reachable(outputs)
    grad = {}
    zero = d.zero

    def scaled(g, f):
        return {k: d.mul(f, v) for k, v in g.items()}

    def combine(ga, gb, sign=1):
        if not gb:
            return ga
        out = dict(ga)
        for k, v in gb.items():
            if k in out:
                out[k] = d.add(out[k], v) if sign > 0 else d.sub(out[k], v)
            else:
                out[k] = v if sign > 0 else d.neg(v)
        return out

    for i in order:
        op = d.op[i]
        if op == ir.CONST:
            g = {}
        elif op == ir.INPUT:
            g = {col_of[i]: d.one} if i in col_of else {}
            for wrt_node, dnode in chain.get(i, ()):
                if wrt_node in col_of:
                    g[col_of[wrt_node]] = dnode
        else:
            a = d.args[i]
            ga = grad[a[0]] if op != ir.SELECT else None
            if op == ir.ADD:
                g = combine(ga, grad[a[1]])
            elif op == ir.SUB:
                g = combine(ga, grad[a[1]], -1)
            elif op == ir.NEG:
                g = {k: d.neg(v) for k, v in ga.items()}
            elif op == ir.MUL:
                gb = grad[a[1]]
                g = combine(scaled(ga, a[1]) if ga else {},
                            scaled(gb, a[0]) if gb else {})
            elif op == ir.DIV:
                gb = grad[a[1]]
                if not ga and not gb:
                    g = {}
                else:
                    inv = d.div(d.one, a[1])
                    g = scaled(ga, inv) if ga else {}
                    if gb:
                        # d(a/b) = da/b - (a/b) db / b
                        g = combine(g, scaled(gb, d.mul(i, inv)), -1)
            elif op == ir.POWI:
                n = a[1]
                g = scaled(ga, d.mul(d.const(n), d.powi(a[0], n - 1))) \
                    if ga else {}
            elif op == ir.POW:
                gb = grad[a[1]]
                g = {}
                if ga:      # b * a**(b-1) da
                    g = scaled(ga, d.mul(a[1], d.pow(
                        a[0], d.sub(a[1], d.one))))
                if gb:      # a**b log(a) db
                    g = combine(g, scaled(gb, d.mul(i, d.unary('log',
                                                              a[0]))))
            elif op in (ir.MAX, ir.MIN):
                gb = grad[a[1]]
                if not ga and not gb:
                    g = {}
                else:
                    diff = d.sub(a[0], a[1]) if op == ir.MAX else \
                        d.sub(a[1], a[0])
                    wa = d.unary('step', diff)       # 1 where a selected
                    wb = d.sub(d.one, wa)
                    g = combine(scaled(ga, wa) if ga else {},
                                scaled(gb, wb) if gb else {})
            elif op == ir.SELECT:
                rel, ca, cb, x, y = a
                gx, gy = grad[x], grad[y]
                g = {k: d.select(rel, ca, cb, gx.get(k, zero),
                                 gy.get(k, zero))
                     for k in set(gx) | set(gy)}
            elif op == ir.ATAN2:
                gb = grad[a[1]]
                if not ga and not gb:
                    g = {}
                else:
                    y, x = a
                    den = d.add(d.mul(x, x), d.mul(y, y))
                    g = combine(scaled(ga, d.div(x, den)) if ga else {},
                                scaled(gb, d.div(y, den)) if gb else {}, -1)
            elif not ga:
                g = {}
            else:
                x = a[0]
                if op == 'sin':
                    f = d.unary('cos', x)
                elif op == 'cos':
                    f = d.neg(d.unary('sin', x))
                elif op == 'tan':
                    f = d.add(d.one, d.mul(i, i))
                elif op == 'exp':
                    f = i
                elif op == 'log':
                    f = d.div(d.one, x)
                elif op == 'sqrt':
                    f = d.div(d.const(0.5), i)
                elif op == 'abs':
                    f = d.unary('sign', x)
                elif op in ('sign', 'step', 'floor', 'ceil'):
                    f = zero
                elif op == 'erf':       # 2/sqrt(pi) exp(-x^2)
                    f = d.mul(d.const(1.1283791670955126),
                              d.unary('exp', d.neg(d.mul(x, x))))
                elif op == 'erfc':
                    f = d.mul(d.const(-1.1283791670955126),
                              d.unary('exp', d.neg(d.mul(x, x))))
                elif op == 'asinh':
                    f = d.div(d.one, d.unary('sqrt', d.add(d.mul(x, x),
                                                           d.one)))
                elif op == 'acosh':
                    f = d.div(d.one, d.unary('sqrt', d.sub(d.mul(x, x),
                                                           d.one)))
                elif op == 'atanh':
                    f = d.div(d.one, d.sub(d.one, d.mul(x, x)))
                elif op == 'asin':
                    f = d.div(d.one, d.unary('sqrt', d.sub(d.one,
                                                          d.mul(x, x))))
                elif op == 'acos':
                    f = d.neg(d.div(d.one, d.unary('sqrt', d.sub(
                        d.one, d.mul(x, x)))))
                elif op == 'atan':
                    f = d.div(d.one, d.add(d.one, d.mul(x, x)))
                elif op == 'sinh':
                    f = d.unary('cosh', x)
                elif op == 'cosh':
                    f = d.unary('sinh', x)
                elif op == 'tanh':
                    f = d.sub(d.one, d.mul(i, i))
                elif op == 'log1p':
                    f = d.div(d.one, d.add(d.one, x))
                elif op == 'expm1':
                    f = d.add(i, d.one)
                elif op == 'log2':      # 1/(x ln 2)
                    f = d.div(d.const(1.4426950408889634), x)
                elif op == 'log10':     # 1/(x ln 10)
                    f = d.div(d.const(0.4342944819032518), x)
                elif op == 'exp2':      # ln 2 * 2^x
                    f = d.mul(d.const(0.6931471805599453), i)
                elif op == 'cbrt':      # 1/(3 cbrt(x)^2)
                    f = d.div(d.one, d.mul(d.const(3.0), d.mul(i, i)))
                else:
                    # tgamma / lgamma: their derivative (digamma) has no C
                    # counterpart -- the reference cannot print it either
                    raise LoweringError('no derivative rule for %s' % op)
                g = scaled(ga, f) if f != zero else {}
        # drop structural zeros that simplification produced
        grad[i] = {k: v for k, v in g.items() if v != zero}

    ncol = len(wrt_inputs)
    return [[grad[o].get(k, zero) for k in range(ncol)] for o in outputs]
