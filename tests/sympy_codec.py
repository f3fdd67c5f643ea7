"""Compact, sharing-preserving JSON codec for SymPy expressions (test
infrastructure).

The gallery fixtures (``tests/golden/gallery_*.npz``) record the *inputs* the
reference's example scripts hand to ``Problem`` -- equations of motion, state
symbols, maps, instance constraints -- next to the outputs the reference
computed for them.  The scripts themselves cannot travel to the GPU box, so
the expressions are stored as data: a post-order table of unique
sub-expressions, ``[head, payload]`` per entry, operands referenced by table
index (``sympy.srepr`` would print a shared sub-tree once per use and nest
past the parser's limits for multibody equations).

``decode(encode(e)) == e`` for everything the reference's printer accepts:
symbols with assumptions, numbers, applied undefined functions
(``dynamicsymbols``), derivatives, relational / Boolean conditions,
``Piecewise``, matrices, and named SymPy functions (looked up in ``sympy``,
``sympy.codegen.cfunctions`` and ``sympy.physics.biomechanics``).
"""
import importlib

import sympy as sm

_MODULES = ('sympy', 'sympy.codegen.cfunctions',
            'sympy.physics.biomechanics', 'sympy.logic.boolalg',
            'sympy.core.relational', 'sympy.functions')


def _lookup(name):
    for mod in _MODULES:
        try:
            m = importlib.import_module(mod)
        except ImportError:
            continue
        if hasattr(m, name):
            return getattr(m, name)
    raise KeyError('no SymPy class named %r' % name)


def _assumptions(obj):
    """What re-creates ``obj`` with equal assumptions: an undefined
    function's ``_kwargs`` / a symbol's ``assumptions0`` (both are the
    deduction-closed sets SymPy compares)."""
    if isinstance(obj, sm.core.function.AppliedUndef):
        a = obj.func._kwargs
    else:
        a = obj.assumptions0
    return {k: bool(v) for k, v in a.items() if v is not None}


def encode(exprs):
    """``exprs``: list of SymPy objects -> JSON-able ``{'table': [...],
    'roots': [...]}``."""
    table, index = [], {}

    def leaf_or_none(e):
        if isinstance(e, sm.Symbol):
            return ['Symbol', [e.name, _assumptions(e)]]
        if isinstance(e, sm.Integer):
            return ['Integer', str(int(e))]
        if isinstance(e, sm.Rational):
            return ['Rational', [str(e.p), str(e.q)]]
        if isinstance(e, sm.Float):
            # exact: mantissa/exponent of the mpf and the precision
            return ['Float', [str(e._mpf_[0]), str(e._mpf_[1]),
                              str(e._mpf_[2]), int(e._prec)]]
        if isinstance(e, (sm.NumberSymbol, sm.core.numbers.ImaginaryUnit)) \
                or e in (sm.nan, sm.oo, -sm.oo, sm.zoo, sm.true, sm.false):
            return ['Singleton', sm.srepr(e)]
        return None

    def visit(root):
        stack = [(root, False)]
        while stack:
            e, ready = stack.pop()
            if e in index:
                continue
            leaf = leaf_or_none(e)
            if leaf is not None:
                index[e] = len(table)
                table.append(leaf)
                continue
            if not ready:
                stack.append((e, True))
                for a in _children(e):
                    if a not in index:
                        stack.append((a, False))
                continue
            index[e] = len(table)
            table.append(_entry(e, index))

    def _children(e):
        if isinstance(e, sm.MatrixBase):
            return list(e)
        if isinstance(e, sm.Derivative):
            return [e.expr] + [v for v, _ in e.variable_count]
        if isinstance(e, sm.Piecewise):
            return [x for pair in e.args for x in pair.args]
        return list(e.args)

    def _entry(e, index):
        if isinstance(e, sm.MatrixBase):
            return ['Matrix', [list(e.shape), [index[x] for x in e]]]
        if isinstance(e, sm.Derivative):
            return ['Derivative', [index[e.expr],
                                   [[index[v], int(c)]
                                    for v, c in e.variable_count]]]
        if isinstance(e, sm.Piecewise):
            return ['Piecewise', [[index[p.args[0]], index[p.args[1]]]
                                  for p in e.args]]
        args = [index[a] for a in e.args]
        if isinstance(e, sm.core.function.AppliedUndef):
            return ['AppliedUndef', [e.func.__name__, _assumptions(e), args]]
        return [type(e).__name__, args]

    roots = []
    for r in exprs:
        r = sm.sympify(r)
        visit(r)
        roots.append(index[r])
    return {'table': table, 'roots': roots, 'sympy': sm.__version__}


def decode(blob):
    """Inverse of :func:`encode`: the list of SymPy objects."""
    out = []
    for head, payload in blob['table']:
        if head == 'Symbol':
            obj = sm.Symbol(payload[0], **payload[1])
        elif head == 'Integer':
            obj = sm.Integer(int(payload))
        elif head == 'Rational':
            obj = sm.Rational(int(payload[0]), int(payload[1]))
        elif head == 'Float':
            sign, man, exp, prec = payload
            man = int(man)
            obj = sm.Float._new((int(sign), man, int(exp),
                                 man.bit_length()), prec, zero=False)
        elif head == 'Singleton':
            obj = sm.sympify(eval(payload, {'__builtins__': {}},
                                  dict(vars(sm))))
        elif head == 'Matrix':
            (r, c), items = payload
            obj = sm.ImmutableDenseMatrix(r, c, [out[k] for k in items])
        elif head == 'Derivative':
            obj = sm.Derivative(out[payload[0]],
                                *[(out[v], c) for v, c in payload[1]])
        elif head == 'Piecewise':
            obj = sm.Piecewise(*[(out[x], out[c]) for x, c in payload],
                               evaluate=False)
        elif head == 'AppliedUndef':
            name, assumptions, args = payload
            obj = sm.Function(name, **assumptions)(*[out[k] for k in args])
        else:
            obj = _lookup(head)(*[out[k] for k in payload])
        out.append(obj)
    return [out[k] for k in blob['roots']]
