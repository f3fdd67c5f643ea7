"""Known-answer cases of the reference's ``TestCreateObjectiveFunction``
(``opty/tests/test_utils.py:67-219``), restated as data: problem, free vector
and the closed-form expected value / gradient, for both the CPU (DAG
interpreter) and the GPU (HIP) tests."""
import numpy as np
import sympy as sym


def cases(N=20, seed=0):
    rng = np.random.default_rng(seed)
    t = sym.symbols('t')
    x, v = [f(t) for f in sym.symbols('x, v', cls=sym.Function)]
    m, c, k = sym.symbols('m, c, k')
    f1, f2 = [f(t) for f in sym.symbols('f1:3', cls=sym.Function)]
    states, inputs, unknowns = [x, v], [f2, f1], [m, c, k]   # to be sorted
    n, q, r = 2, 2, 3
    xv, vv, f1v, f2v = (rng.random(N) for _ in range(4))
    mv, cv, kv = rng.random(3)
    free = np.hstack((xv, vv, f1v, f2v, cv, kv, mv))   # params sorted c, k, m
    z = np.zeros
    out = []
    out.append(dict(
        name='backward_single_state', expr=sym.Integral(x**2, t), h=0.5,
        method='backward euler', args=(states, inputs, unknowns), free=free,
        value=0.5*(xv[1:]**2).sum(),
        grad=np.hstack((0, 0.5*2*xv[1:], z(N*(1 + q) + r)))))
    out.append(dict(
        name='backward_single_input', expr=sym.Integral(f1**2, t), h=1.0,
        method='backward euler', args=(states, inputs, unknowns), free=free,
        value=(f1v[1:]**2).sum(),
        grad=np.hstack((z(N*n + 1), 2*f1v[1:], z(N + r)))))
    out.append(dict(
        name='backward_single_unknown', expr=m**2, h=0.3,
        method='backward euler', args=(states, inputs, unknowns), free=free,
        value=mv**2, grad=np.hstack((z(N*(n + q) + 2), 2*mv))))
    allexpr = (sym.Integral(x**2 + m**2, t) + sym.Integral(c**2*f2**2, t) +
               sym.sin(k)**2)
    out.append(dict(
        name='backward_all', expr=allexpr, h=0.3, method='backward euler',
        args=(states, inputs, unknowns), free=free,
        value=0.3*((xv[1:]**2).sum() + (N - 1)*mv**2 +
                   (cv**2*f2v[1:]**2).sum()) + np.sin(kv)**2,
        grad=np.hstack((0, 0.3*2*xv[1:], z(N*2 + 1), 0.3*2*cv**2*f2v[1:],
                        0.3*2*cv*(f2v[1:]**2).sum(),
                        2*np.sin(kv)*np.cos(kv), 0.3*(N - 1)*2*mv))))
    x_mid, f2_mid = (xv[1:] + xv[:-1])/2, (f2v[1:] + f2v[:-1])/2
    out.append(dict(
        name='midpoint_all', expr=allexpr, h=0.3, method='midpoint',
        args=(states, inputs, unknowns), free=free,
        value=0.3*((x_mid**2).sum() + (N - 1)*mv**2 +
                   (cv**2*f2_mid**2).sum()) + np.sin(kv)**2,
        grad=np.hstack((0.3*xv[0], 0.3*2*xv[1:-1], 0.3*xv[-1], z(N*2),
                        0.3*cv**2*f2v[0], 0.3*2*cv**2*f2v[1:-1],
                        0.3*cv**2*f2v[-1], 0.3*2*cv*(f2_mid**2).sum(),
                        2*np.sin(kv)*np.cos(kv), 0.3*(N - 1)*2*mv))))
    out.append(dict(
        name='no_states', expr=sym.Integral(f1**2, t), h=1.0,
        method='backward euler', args=([], inputs, unknowns),
        free=free[n*N:], value=(f1v[1:]**2).sum(),
        grad=np.hstack((0, 2*f1v[1:], z(N + r)))))
    out.append(dict(
        name='no_inputs', expr=sym.Integral(x**2, t), h=1.0,
        method='backward euler', args=(states, [], unknowns),
        free=np.hstack((free[:n*N], free[-r:])), value=(xv[1:]**2).sum(),
        grad=np.hstack((0, 2*xv[1:], z(N + r)))))
    out.append(dict(
        name='no_unknowns', expr=sym.Integral(x**2, t), h=1.0,
        method='backward euler', args=(states, inputs, []),
        free=free[:-r], value=(xv[1:]**2).sum(),
        grad=np.hstack((0, 2*xv[1:], z(N*(n - 1 + q))))))
    return t, out


def reference_cases():
    """Problems whose objective value / gradient were recorded from the
    reference's own ``create_objective_function``
    (``opty/utils.py:329-470``) by ``tests/golden/_gen/make_golden.py
    objective`` -> ``tests/golden/objective.npz``.  Inputs are the
    deterministic ``problems.make_free`` recipe (seed per case)."""
    t = sym.symbols('t')
    x, v, u = [f(t) for f in sym.symbols('x, v, u', cls=sym.Function)]
    f1, f2 = [f(t) for f in sym.symbols('f1:3', cls=sym.Function)]
    m, c, k, p = sym.symbols('m, c, k, p')
    allexpr = (sym.Integral(x**2 + m**2, t) + sym.Integral(c**2*f2**2, t) +
               sym.sin(k)**2)
    trig = sym.Integral(p*u**2 + sym.cos(x)*v**2, t) + 3*p**2
    out = []
    for method in ('backward euler', 'midpoint'):
        tag = 'be' if method == 'backward euler' else 'mid'
        out.append(dict(name='all_' + tag, expr=allexpr, method=method,
                        N=20, h=0.3, seed=31,
                        args=([x, v], [f2, f1], [m, c, k])))
        out.append(dict(name='trig_' + tag, expr=trig, method=method,
                        N=1001, h=0.01, seed=32, args=([x, v], [u], [p])))
    # no unknown parameters: the swing-up's minimum-effort objective
    out.append(dict(name='effort_be', expr=sym.Integral(u**2, t),
                    method='backward euler', N=777, h=10.0/776, seed=33,
                    args=([x, v], [u], [])))
    out.append(dict(name='states_only_mid',
                    expr=sym.Integral(sym.exp(-x)*v**2 + sym.sqrt(1 + x**2),
                                      t),
                    method='midpoint', N=130, h=0.05, seed=34,
                    args=([x, v], [], [])))
    for case in out:
        states, inputs, unknowns = case['args']
        case['num_free'] = (len(states) + len(inputs))*case['N'] + \
            len(unknowns)
    return t, out
