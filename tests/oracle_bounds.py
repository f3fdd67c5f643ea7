"""Term magnitudes of the ORACLE's own expressions (test infrastructure).

The parity tests widen an entry's tolerance beyond 1e-10 relative only by what
cancellation inside that entry justifies, measured by a running rounding-error
bound of the PRODUCT's expression DAG (``dag_interp.error_bounds``).  A
numerically poor rewrite in the product could therefore widen its own
tolerance.  This module computes, from the SymPy expressions the oracle (and
the reference) evaluates -- the discretised equations and their
``Matrix.jacobian`` -- an independent measure per entry: the value the
expression takes when every sum adds the ABSOLUTE values of its terms,

    mag(a + b) = mag(a) + mag(b),   mag(a*b) = mag(a)*mag(b),
    mag(f(a)) = |f(a)| + |f'(a)|*mag(a),   mag(x) = |x|,

i.e. the size of the terms that cancel.  ``2**-53 * mag`` is the scale of the
rounding error ANY evaluation order of that expression commits;
``tests/test_oracle_golden.py`` holds the product-side bounds to a fixed
multiple of it."""
import numpy as np
import sympy as sm

#: lambdify name spaces: what the scipy / numpy printers do not know of the
#: C99 printer's function table (sympy.codegen.cfunctions)
_MODULES = [{'Cbrt': np.cbrt, 'hypot': np.hypot, 'exp2': np.exp2,
             'log2': np.log2, 'log10': np.log10, 'log1p': np.log1p,
             'expm1': np.expm1, 'fma': lambda a, b, c: a*b + c},
            'scipy', 'numpy']


class _Evaluator(object):

    def __init__(self, values):
        self.values = values            # Symbol -> ndarray / float
        self.memo = {}

    def __call__(self, e):
        key = e
        hit = self.memo.get(key)
        if hit is None:
            hit = self.memo[key] = self._eval(e)
        return hit

    def _eval(self, e):
        if e.is_Number or isinstance(e, sm.NumberSymbol):
            v = float(e)
            return v, abs(v)
        if e.is_Symbol:
            v = self.values[e]
            return v, np.abs(v)
        if isinstance(e, sm.Add):
            parts = [self(a) for a in e.args]
            return sum(p[0] for p in parts), sum(p[1] for p in parts)
        if isinstance(e, sm.Mul):
            v, m = 1.0, 1.0
            for a in e.args:
                pv, pm = self(a)
                v, m = v*pv, m*pm
            return v, m
        if isinstance(e, sm.Pow):
            bv, bm = self(e.base)
            if e.exp.is_Number:
                p = float(e.exp)
                v = np.power(bv, p)
                with np.errstate(all='ignore'):
                    amp = np.where(np.abs(bv) > 0, bm/np.abs(bv), 1.0)
                return v, np.abs(v)*np.power(amp, abs(p))
            pv, pm = self(e.exp)
            v = np.power(bv, pv)
            return v, np.abs(v)*(1.0 + np.abs(pv)*bm/np.maximum(
                np.abs(bv), 1e-300) + np.abs(np.log(np.abs(bv) + 1e-300))*pm)
        if isinstance(e, sm.Piecewise):
            vals = [self(x) for x, _ in e.args]
            conds = [np.asarray(self._cond(c)) for _, c in e.args]
            shape = np.broadcast(*(conds + [np.asarray(x[0]) for x in vals] +
                                   [np.asarray(x[1]) for x in vals])).shape
            conds = [np.broadcast_to(c, shape) for c in conds]
            v = np.select(conds, [np.broadcast_to(x[0], shape) for x in vals])
            m = np.select(conds, [np.broadcast_to(x[1], shape) for x in vals])
            return v, m
        if isinstance(e, (sm.Max, sm.Min)):
            parts = [self(a) for a in e.args]
            f = np.maximum if isinstance(e, sm.Max) else np.minimum
            v = parts[0][0]
            for p in parts[1:]:
                v = f(v, p[0])
            return v, sum(p[1] for p in parts)
        if isinstance(e, sm.UnevaluatedExpr):
            return self(e.args[0])
        if isinstance(e, sm.Function):
            args = [self(a) for a in e.args]
            fn = sm.lambdify(sm.symbols('x0:%d' % len(args)),
                             e.func(*sm.symbols('x0:%d' % len(args))),
                             _MODULES)
            with np.errstate(all='ignore'):
                v = fn(*[a[0] for a in args])
                m = np.abs(v)
                xs = sm.symbols('x0:%d' % len(args))
                for k, a in enumerate(args):
                    try:
                        d = sm.lambdify(xs, e.func(*xs).diff(xs[k]),
                                        _MODULES)
                        dv = np.abs(d(*[b[0] for b in args]))
                    except Exception:
                        dv = 1.0
                    m = m + np.where(np.isfinite(dv), dv, 1.0)*a[1]
            return v, m
        raise NotImplementedError(type(e))

    def _cond(self, c):
        if c is sm.true or c is True:
            return np.array(True)
        if c is sm.false or c is False:
            return np.array(False)
        if isinstance(c, sm.And):
            out = self._cond(c.args[0])
            for a in c.args[1:]:
                out = out & self._cond(a)
            return out
        if isinstance(c, sm.Or):
            out = self._cond(c.args[0])
            for a in c.args[1:]:
                out = out | self._cond(a)
            return out
        if isinstance(c, sm.Not):
            return ~self._cond(c.args[0])
        ops = {sm.Lt: np.less, sm.Le: np.less_equal, sm.Gt: np.greater,
               sm.Ge: np.greater_equal, sm.Eq: np.equal, sm.Ne: np.not_equal}
        return ops[type(c)](self(c.lhs)[0], self(c.rhs)[0])


def magnitudes(orc, free):
    """``(con_mag (M*(N-1),), jac_mag (P*(N-1),))`` in the layouts of
    ``constraints(free)`` / ``jacobian(free)`` (collocation part only) for an
    ``OracleCollocator``; ``orc._gen_jac()`` must have run (it does inside
    ``generate_jacobian_function``)."""
    from oracle.collocation_oracle import split_free
    free = np.asarray(free, dtype=float)
    states, spec, consts, h = split_free(free, orc.n, orc.q, orc.N,
                                         orc.variable_duration)
    if not orc.variable_duration:
        h = orc.node_time_interval
    all_spec = orc._merge(orc.trajectories, orc.known_trajectory_map, spec,
                          True, free)
    all_const = orc._merge(orc.parameters, orc.known_parameter_map, consts,
                           False, free)
    vec = orc._node_vectors(states, all_spec if orc.m else np.zeros((0,
                                                                       orc.N)))
    args = [orc.func_repl.get(a, a) for a in orc.args]
    nvec = len(vec)
    values = {s: np.asarray(v, dtype=float) for s, v in zip(args[:nvec], vec)}
    values.update({s: float(v) for s, v in zip(args[nvec:],
                                               list(all_const) + [h])})
    ev = _Evaluator(values)
    ones = np.ones(orc.N - 1)
    eom = orc.discrete_eom.xreplace(orc.func_repl)
    con = np.stack([ev(e)[1]*ones for e in eom])                # (M, N-1)
    jac = np.stack([ev(e)[1]*ones for e in orc.symbolic_jacobian], axis=1)
    return con.ravel(), jac.ravel()
